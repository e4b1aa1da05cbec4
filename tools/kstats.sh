#!/bin/bash
# rocprofv3 kernel stats of one bench.py command -> gpurun_out/profiles_out/<tag>_kernel_stats.csv
#   bash tools/kstats.sh r04_bench_ref --workload ref --steps 3 --warmup 2 --no-cpu-baseline --no-alt
TAG=$1; shift
ROOT=$(pwd)
export TMPDIR=/tmp
mkdir -p gpurun_out/profiles_out
rm -rf gpurun_out/prof_$TAG
cd /tmp
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_$TAG -o run --output-format csv -- python3 $ROOT/bench.py "$@" > $ROOT/gpurun_out/prof_$TAG.log 2>&1
cd $ROOT
cp $(find gpurun_out/prof_$TAG -name '*kernel_stats.csv' | head -1) gpurun_out/profiles_out/${TAG}_kernel_stats.csv
grep '"metric"' gpurun_out/prof_$TAG.log | tail -1 > gpurun_out/profiles_out/${TAG}_line_under_rocprof.json
python3 - <<PY
import csv
rows = list(csv.DictReader(open("gpurun_out/profiles_out/${TAG}_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", round(tot / 1e6, 2), "launches", sum(int(r["Calls"]) for r in rows))
for r in rows[:24]:
    print(f"{r['Name'][:96]:96s} n={int(r['Calls']):5d} avg_us={float(r['AverageNs'])/1e3:9.1f} ms={float(r['TotalDurationNs'])/1e6:8.2f} {float(r['Percentage']):5.1f}%")
PY
