#!/bin/bash
# fp32 parity mode, the plane-kernel engine (UMR_X3_ALL=1, default) against round 3's form (UMR_X3_ALL=0) on ONE box:
# the reference recipe's train step, the cfg5 sweep, and a rocprofv3 kernel-stats pass of the ref step.
ROOT=$(pwd)
export TMPDIR=/tmp
mkdir -p gpurun_out
for v in 1 0; do
  UMR_X3_ALL=$v python bench.py --workload ref --dtype fp32 --no-cpu-baseline --no-alt --steps 8 > gpurun_out/r4_ref_fp32_x3all$v.json 2> gpurun_out/r4_ref_fp32_x3all$v.err
  UMR_X3_ALL=$v python bench.py --workload cfg5 --no-cpu-baseline --no-alt --steps 2 --warmup 1 > gpurun_out/r4_cfg5_fp32_x3all$v.json 2> gpurun_out/r4_cfg5_fp32_x3all$v.err
done
cd /tmp
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_r4_ref_fp32 -o run --output-format csv -- python3 $ROOT/bench.py --workload ref --dtype fp32 --steps 3 --warmup 2 --no-cpu-baseline --no-alt --graphs off > $ROOT/gpurun_out/prof_r4_ref_fp32.log 2>&1
cd $ROOT
cp $(find gpurun_out/prof_r4_ref_fp32 -name '*kernel_stats.csv' | head -1) gpurun_out/r4_ref_fp32_kernel_stats.csv
for f in gpurun_out/r4_ref_fp32_x3all*.json gpurun_out/r4_cfg5_fp32_x3all*.json; do python -c "
import json,sys
r=json.load(open('$f')); print('$f', round(r['value'],3), round(r['ms_per_step'],2), r['roofline']['achieved'])"; done
