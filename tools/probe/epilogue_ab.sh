#!/bin/bash
# same-box A/B of the persistent 256x256 kernel's fast-class epilogue: unmore_amd/lib/libumr_base.so (built from the previous commit's
# gemm_nt256p.hip) against the working tree's library.  GPU box, repo root.
for i in 1 2; do
echo "== base"; UMR_LIB=unmore_amd/lib/libumr_base.so timeout -k 10 200 python tools/probe/red_bench.py 2>&1 | grep -E "ms "
echo "== new";  timeout -k 10 200 python tools/probe/red_bench.py 2>&1 | grep -E "ms "
done
echo "== base"; UMR_LIB=unmore_amd/lib/libumr_base.so timeout -k 10 200 python tools/kbench.py 64 2>&1 | grep -E "conv3x3|1x1"
echo "== new";  timeout -k 10 200 python tools/kbench.py 64 2>&1 | grep -E "conv3x3|1x1"
echo "== stamps (new)"
UMR_LIB=unmore_amd/lib/libumr_ts.so timeout -k 10 200 python tools/probe/ts_probe.py red 2>&1 | grep -E "aux mode|tile [345]:" | sed "s/ | k-tile0.*//"
