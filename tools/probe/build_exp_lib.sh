#!/bin/bash
# experimental copy of the library: ONE source rebuilt with extra -D flags -> unmore_amd/lib/libumr_exp.so  (A/B via UMR_LIB)
#   bash tools/probe/build_exp_lib.sh gemm_nt256p.hip -DUMR_EXP_PRIO_MODE=0
set -e
SRC=$1; shift
cd "$(dirname "$0")/../../unmore_amd/csrc"
make -j8 > /dev/null
EXTRA=""; [ "$SRC" = gemm_nt256p.hip ] && EXTRA="-fno-honor-nans"
[ "$SRC" = attention.hip ] && EXTRA="-mllvm -amdgpu-mfma-vgpr-form=1 -fno-honor-nans"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result $EXTRA "$@" -c $SRC -o build/exp_tmp.o
OBJS=$(for f in *.hip; do [ "$f" != "$SRC" ] && echo build/${f%.hip}.o; done)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libumr_exp.so $OBJS build/exp_tmp.o
rm -f build/exp_tmp.o
echo built unmore_amd/lib/libumr_exp.so $SRC "$@"
