#!/bin/bash
# same-box A/B: epilogue of the persistent 256x256 kernel with LDS-staged 16-byte copies (default) against stores straight from the
# fragment layout (v_permlane16_swap pairs -> 16 bytes per lane).  GPU box, repo root:  bash tools/probe/direct_store_ab.sh [-DUMR_EXP_DIRECT_STORE=2]
FLAG=${1:--DUMR_EXP_DIRECT_STORE=1}
bash tools/probe/build_exp_lib.sh gemm_nt256p.hip $FLAG > /dev/null || exit 1
for i in 1 2; do
echo "== base"; timeout -k 10 200 python tools/probe/red_bench.py 2>&1 | grep -E "ms "
echo "== exp $FLAG"; UMR_LIB=unmore_amd/lib/libumr_exp.so timeout -k 10 200 python tools/probe/red_bench.py 2>&1 | grep -E "ms "
done
echo "== base"; timeout -k 10 200 python tools/kbench.py 64 2>&1 | grep -E "conv3x3 512->512 fwd|1x1"
echo "== exp";  UMR_LIB=unmore_amd/lib/libumr_exp.so timeout -k 10 200 python tools/kbench.py 64 2>&1 | grep -E "conv3x3 512->512 fwd|1x1"
UMR_LIB=unmore_amd/lib/libumr_exp.so timeout -k 10 600 python -m pytest tests/test_gemm_gpu.py -x -q 2>&1 | tail -2
