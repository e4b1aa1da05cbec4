#!/bin/bash
# where the tile time goes with the direct-store epilogue (s_memtime stamps): bash tools/probe/direct_store_ts.sh "<flags>" ...
for FL in "$@"; do
bash tools/probe/build_exp_lib.sh gemm_nt256p.hip -DUMR_NT256P_TIMESTAMPS $FL > /dev/null || exit 1
echo "==== $FL"
UMR_LIB=unmore_amd/lib/libumr_exp.so timeout -k 10 200 python tools/probe/ts_probe.py red 2>&1 | grep -E "aux mode|tile [345]:" | sed "s/ | k-tile0.*//"
bash tools/probe/build_exp_lib.sh gemm_nt256p.hip $FL > /dev/null || exit 1
UMR_LIB=unmore_amd/lib/libumr_exp.so timeout -k 10 200 python tools/probe/red_bench.py 2>&1 | grep -E "ms "
done
