#!/bin/bash
# same-box A/B of the bf16 attention kernels' occupancy (GPU box, repo root): forward at 3 (default) vs 4 waves per SIMD,
# dK/dV at 2 (default) vs 3
set -e
for i in 1 2; do
echo "== default"; timeout -k 10 120 python tools/attn_bench.py 2>&1 | grep attention
bash tools/probe/build_exp_lib.sh attention.hip -DUMR_ATTN_FWD_MIN_WAVES=4 > /dev/null
echo "== fwd 4 waves/SIMD"; UMR_LIB=unmore_amd/lib/libumr_exp.so timeout -k 10 120 python tools/attn_bench.py 2>&1 | grep attention
bash tools/probe/build_exp_lib.sh attention.hip -DUMR_ATTN_DKV_MIN_WAVES=3 > /dev/null
echo "== dkv 3 waves/SIMD"; UMR_LIB=unmore_amd/lib/libumr_exp.so timeout -k 10 120 python tools/attn_bench.py 2>&1 | grep attention
done
