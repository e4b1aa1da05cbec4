# same-box A/B of experimental build flags of gemm_nt256p.hip against the default library (GPU box, repo root):
#   bash tools/probe/exp_ab.sh -DUMR_EXP_PH2_ALL
bash tools/probe/build_exp_lib.sh gemm_nt256p.hip "$@" > /dev/null
for i in 1 2; do
echo "base:"; timeout -k 10 200 python tools/kbench.py 64 2>&1 | grep -E "conv3x3 512->512 fwd"
echo "exp :"; UMR_LIB=unmore_amd/lib/libumr_exp.so timeout -k 10 200 python tools/kbench.py 64 2>&1 | grep -E "conv3x3 512->512 fwd"
done
UMR_LIB=unmore_amd/lib/libumr_exp.so timeout -k 10 300 python -m pytest tests/test_gemm_gpu.py -x -q -k "large_tile or transformer" 2>&1 | tail -1
