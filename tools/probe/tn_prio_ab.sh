#!/bin/bash
# plain weight-gradient GEMMs on the 256x256 TN kernel: priority flips around the MFMA clusters (product) against a static priority for
# waves 4-7 (the conv's form).   bash tools/probe/tn_prio_ab.sh   (GPU box, repo root)
set -e
run() { echo "== $1"; shift
  "$@" python tools/kbench.py 64 2>&1 | grep -E "wgrad"
  "$@" python tools/vit_block_bench.py 2>&1 | grep -E "wgrad|backward|block " 
  "$@" python tools/x3_bench.py 2>&1 | grep -E "^TN"; }
run "product library" env
bash tools/probe/build_exp_lib.sh gemm_tn256.hip -DUMR_EXP_TN_PLAIN_PRIO=1 > /dev/null
run "static priority for waves 4-7" env UMR_LIB=$(pwd)/unmore_amd/lib/libumr_exp.so
bash tools/probe/build_exp_lib.sh gemm_tn256.hip -DUMR_EXP_TN_PLAIN_PRIO=3 > /dev/null
run "no priorities" env UMR_LIB=$(pwd)/unmore_amd/lib/libumr_exp.so
