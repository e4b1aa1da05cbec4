#!/bin/bash
# one FETCH_SIZE pass of a bench.py command, per (kernel, grid): launches, avg fetched MB per launch (x2 gfx950 correction), avg us
#   bash tools/probe/fetch_pass.sh --workload ref --graphs off --steps 3 --warmup 2 --no-cpu-baseline --no-alt
ROOT=$(pwd); export TMPDIR=/tmp; rm -rf /tmp/fetch_pass; cd /tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/fetch_pass -o run --output-format csv -- python3 $ROOT/bench.py "$@" > /tmp/fetch_pass.log 2>&1
python3 - <<'PY'
import csv, glob, collections, re
acc = collections.defaultdict(lambda: [0, 0.0, 0.0])
for f in glob.glob("/tmp/fetch_pass/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") != "FETCH_SIZE":
            continue
        n = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"])[:70]
        k = (n, int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"])))
        acc[k][0] += 1
        acc[k][1] += float(r["Counter_Value"])
        acc[k][2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) if "End_Timestamp" in r else 0
rows = sorted(acc.items(), key=lambda kv: -kv[1][1])[:40]
for (n, g), (c, v, t) in rows:
    print(f"{n:70s} wg {g:6d} n {c:5d} fetched/launch {2 * 1024 * v / c / 1e6:9.2f} MB  total {2 * 1024 * v / 1e9:7.2f} GB")
PY
