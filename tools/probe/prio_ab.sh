# same-box A/B of the MFMA-cluster priority policy (GPU box, repo root):  bash tools/probe/prio_ab.sh {nt|tn}
K=${1:-nt}
if [ $K = nt ]; then SRC=gemm_nt256p.hip; D=UMR_EXP_PRIO_MODE; PAT='conv3x3 512->512 fwd'; else SRC=gemm_tn256.hip; D=UMR_EXP_TN_PRIO_MODE; PAT='TN'; fi
for m in 0 1 3; do bash tools/probe/build_exp_lib.sh $SRC -D$D=$m > /dev/null; cp unmore_amd/lib/libumr_exp.so unmore_amd/lib/libumr_exp$m.so; done
for i in 1 2; do for m in 0 1 3; do
  echo "mode $m:"; UMR_LIB=unmore_amd/lib/libumr_exp$m.so timeout -k 10 200 python tools/kbench.py 64 2>&1 | grep -E "$PAT"
done; done
