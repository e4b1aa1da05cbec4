#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of one kernel shape:  bash tools/kfetch.sh {convnt|convtn|gemm1x1} [B] [TAG]   (GPU box, repo root)
set -e
W=${1:-convnt}; B=${2:-64}; TAG=${3:-x}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/kfetch_${W}_$TAG
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/f -o run --output-format csv -- python3 $ROOT/tools/kprof.py $W $B > $OUT/f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/w -o run --output-format csv -- python3 $ROOT/tools/kprof.py $W $B > $OUT/w.log 2>&1
cd $ROOT
python3 - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
for name, sub, mul in (("FETCH_SIZE", "f", 2048.0), ("WRITE_SIZE", "w", 1024.0)):
    best = {}
    for f in glob.glob(out + "/" + sub + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != name or "gemm" not in r["Kernel_Name"]:
                continue
            k = r["Kernel_Name"][:70]
            ms = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
            best.setdefault(k, []).append((float(r["Counter_Value"]) * mul / 1e9, ms))
    for k, v in best.items():
        print(name, k, "GB/launch %.2f" % (sum(x[0] for x in v) / len(v)), "ms %.3f" % (sum(x[1] for x in v) / len(v)), "n", len(v))
PY
