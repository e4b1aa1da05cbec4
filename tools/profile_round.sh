#!/bin/bash
# Round profile: rocprofv3 kernel stats + separate FETCH_SIZE / WRITE_SIZE / MfmaUtil PMC passes of bench.py, condensed into profiles/.
# usage (on the GPU box, from the repo root):  bash tools/profile_round.sh r01
set -e
TAG=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT $ROOT/profiles
export TMPDIR=/tmp
ARGS="--steps 3 --warmup 2 --no-cpu-baseline --no-alt"
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o run --output-format csv -- python3 $ROOT/bench.py $ARGS > $OUT/stats.log 2>&1
echo "stats pass done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o run --output-format csv -- python3 $ROOT/bench.py $ARGS > $OUT/fetch.log 2>&1
echo "fetch pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o run --output-format csv -- python3 $ROOT/bench.py $ARGS > $OUT/write.log 2>&1
echo "write pass done"
rocprofv3 --kernel-trace --pmc MfmaUtil -d $OUT/mfma -o run --output-format csv -- python3 $ROOT/bench.py $ARGS > $OUT/mfma.log 2>&1
echo "mfma pass done"
cd $ROOT
cp $(find $OUT/stats -name '*kernel_stats.csv' | head -1) profiles/${TAG}_bench_cfg2_kernel_stats.csv
python3 tools/pmc_summary.py --stats $OUT/stats --fetch $OUT/fetch --write $OUT/write --mfma $OUT/mfma \
    --out profiles/${TAG}_pmc_traffic.json --command "python bench.py $ARGS"
mkdir -p gpurun_out/profiles_$TAG && cp profiles/${TAG}_* gpurun_out/profiles_$TAG/
grep '"metric"' $OUT/stats.log | tail -1 > gpurun_out/profiles_$TAG/bench_line_under_rocprof.json || true
