#!/bin/bash
# same-box A/B of the plane-pair order in the X3 K loop (GPU box, repo root)
set -e
bash tools/probe/build_exp_lib.sh gemm_nt256p.hip -DUMR_EXP_X3_ORDER_A > /dev/null
for i in 1 2; do
timeout -k 10 120 python tools/energy_probe.py --kernel conv_x3 --seconds 4 --tag default
UMR_LIB=unmore_amd/lib/libumr_exp.so timeout -k 10 120 python tools/energy_probe.py --kernel conv_x3 --seconds 4 --tag A_planes_consecutive
done
UMR_LIB=unmore_amd/lib/libumr_exp.so timeout -k 10 200 python -m pytest tests/test_gemm_gpu.py -x -q -k "x3" 2>&1 | tail -1
