#!/bin/bash
# same-box A/B of the GELU epilogue class: libumr_base.so (gemm_nt256p.hip of the round's start) against the working tree
for i in 1 2; do
echo "== base"; UMR_LIB=unmore_amd/lib/libumr_base.so timeout -k 10 200 python tools/probe/gelu_bench.py 2>&1 | grep "us "
echo "== new";  timeout -k 10 200 python tools/probe/gelu_bench.py 2>&1 | grep "us "
done
