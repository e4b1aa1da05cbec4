#!/bin/bash
# Round profile: rocprofv3 kernel stats + separate FETCH_SIZE / WRITE_SIZE / MfmaUtil PMC passes of bench.py, condensed into profiles/.
# usage (on the GPU box, from the repo root):  bash tools/profile_round.sh r02 [cfg2|cfg4|cfg5|cfg1] [stats-only]
# (--pmc passes are separate runs with --kernel-trace only: gpurun refuses --pmc combined with the other trace domains)
set -e
TAG=${1:-r02}
WL=${2:-cfg2}
MODE=${3:-full}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_${TAG}_${WL}
rm -rf $OUT
mkdir -p $OUT $ROOT/profiles
export TMPDIR=/tmp
ARGS="--workload $WL --steps 3 --warmup 2 --no-cpu-baseline --no-alt"
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o run --output-format csv -- python3 $ROOT/bench.py $ARGS > $OUT/stats.log 2>&1
echo "stats pass done"
cd $ROOT
cp $(find $OUT/stats -name '*kernel_stats.csv' | head -1) profiles/${TAG}_bench_${WL}_kernel_stats.csv
grep '"metric"' $OUT/stats.log | tail -1 > profiles/${TAG}_bench_${WL}_line_under_rocprof.json || true
mkdir -p gpurun_out/profiles_out && cp profiles/${TAG}_bench_${WL}_* gpurun_out/profiles_out/   # profiles/ itself does not travel back
if [ "$MODE" = "stats-only" ]; then exit 0; fi
cd /tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o run --output-format csv -- python3 $ROOT/bench.py $ARGS > $OUT/fetch.log 2>&1
echo "fetch pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o run --output-format csv -- python3 $ROOT/bench.py $ARGS > $OUT/write.log 2>&1
echo "write pass done"
rocprofv3 --kernel-trace --pmc MfmaUtil -d $OUT/mfma -o run --output-format csv -- python3 $ROOT/bench.py $ARGS > $OUT/mfma.log 2>&1
echo "mfma pass done"
cd $ROOT
SUFFIX=""; if [ "$WL" != "cfg2" ]; then SUFFIX="_$WL"; fi
python3 tools/pmc_summary.py --stats $OUT/stats --fetch $OUT/fetch --write $OUT/write --mfma $OUT/mfma \
    --out profiles/${TAG}_pmc_traffic${SUFFIX}.json --command "python bench.py $ARGS"
cp profiles/${TAG}_pmc_traffic${SUFFIX}.json gpurun_out/profiles_out/
