#!/bin/bash
# rocprofv3 kernel-stats pass of one bench.py command; prints the top kernels.  usage (GPU box, repo root): bash tools/stats_pass.sh TAG [bench args...]
set -e
TAG=${1:-x}; shift || true
ROOT=$(pwd); OUT=$ROOT/gpurun_out/stats_$TAG
rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/s -o run --output-format csv -- python3 $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-alt "$@" > $OUT/bench.log 2>&1
cd $ROOT
F=$(find $OUT/s -name '*kernel_stats.csv' | head -1)
cp $F $OUT/kernel_stats.csv
python3 - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms (5 steps incl. warm-up):", tot / 1e6)
for r in rows[:28]:
    print("%-110s calls %5s avg_ms %9.3f total_ms %9.2f %5.1f%%" % (r["Name"][:110], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e6, float(r["Percentage"])))
PY
grep '"metric"' $OUT/bench.log | tail -1 | cut -c1-400
