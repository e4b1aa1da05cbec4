#!/bin/bash
# Does the XCD-aware tile order of the persistent GEMM buy L2 sharing of the A panel between the N-tiles of an M-tile?
# FETCH_SIZE of the masked 1024 -> 512 data gradient and of the 3x3 conv with the product library and with a build whose
# workgroups take tiles in blockIdx order (-DUMR_EXP_NO_XCD_REMAP).   bash tools/probe/xcd_remap_fetch.sh   (GPU box, repo root)
set -e
bash tools/probe/build_exp_lib.sh gemm_nt256p.hip -DUMR_EXP_NO_XCD_REMAP
for W in dgrad1x1 convnt; do
  echo "== $W, product library"; bash tools/kfetch.sh $W 64 prod
  echo "== $W, tiles in blockIdx order"; UMR_LIB=$(pwd)/unmore_amd/lib/libumr_exp.so bash tools/kfetch.sh $W 64 noremap
done
rm -rf gpurun_out/kfetch_*
