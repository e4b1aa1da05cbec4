#!/bin/bash
# instrumented copy of the library for tools/probe/ts_probe.py (only gemm_nt256p.hip differs)
set -e
cd "$(dirname "$0")/../../unmore_amd/csrc"
make -j8 > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -fno-honor-nans -DUMR_NT256P_TIMESTAMPS -c gemm_nt256p.hip -o build/gemm_nt256p_ts.o
OBJS=$(for f in *.hip; do [ "$f" != gemm_nt256p.hip ] && echo build/${f%.hip}.o; done)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libumr_ts.so $OBJS build/gemm_nt256p_ts.o
rm -f build/gemm_nt256p_ts.o
echo built unmore_amd/lib/libumr_ts.so
