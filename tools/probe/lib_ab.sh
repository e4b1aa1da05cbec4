#!/bin/bash
# same-box A/B of two builds of the library on the reference recipe's step:  bash tools/probe/lib_ab.sh <base .so> [workload]
BASE=$1; WL=${2:-ref}
for i in 1 2 3; do
for L in $BASE unmore_amd/lib/libumr.so; do
UMR_LIB=$L python bench.py --workload $WL --steps 20 --warmup 5 --no-cpu-baseline --no-alt 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$L', round(d['value'],1), 'images/s', round(d['ms_per_step'],2), 'ms/step')"
done; done
