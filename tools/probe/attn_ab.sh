#!/bin/bash
# same-box A/B of attention kernel builds: libraries named on the command line (paths under unmore_amd/lib), alternating, 3 rounds,
# at the cfg2 and cfg4 shapes.   bash tools/probe/attn_ab.sh libumr_attn_r04.so libumr_attn_v2.so libumr.so
ROOT=$(pwd)
for shape in 64,577,12 16,1370,16; do
  echo "== shape (B,N,heads) = $shape"
  for round in 1 2 3; do
    for lib in "$@"; do
      printf "%-22s " $lib
      UMR_LIB=$ROOT/unmore_amd/lib/$lib ATTN_SHAPE=$shape ATTN_ONLY=1 python tools/attn_bench.py 2>/dev/null | tr '\n' ' '
      echo
    done
  done
done
