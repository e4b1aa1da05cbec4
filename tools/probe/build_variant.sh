#!/bin/bash
# a copy of the library with one source rebuilt under extra -D flags:
#   bash tools/probe/build_variant.sh linear_head.hip u8 -DLH_WEIGHT_UNROLL=8   ->  unmore_amd/lib/libumr_u8.so  (use with UMR_LIB=...)
set -e
SRC=$1; SUF=$2; shift 2
cd "$(dirname "$0")/../../unmore_amd/csrc"
make -j8 > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -fno-honor-nans "$@" -c $SRC -o build/${SRC%.hip}_$SUF.o
OBJS=$(for f in *.hip; do [ "$f" != $SRC ] && echo build/${f%.hip}.o; done)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libumr_$SUF.so $OBJS build/${SRC%.hip}_$SUF.o
rm -f build/${SRC%.hip}_$SUF.o
echo built unmore_amd/lib/libumr_$SUF.so
