#!/bin/bash
# arbitrary PMC counters of one kernel shape:  bash tools/kpmc.sh {convnt|convtn|gemm1x1} B "CTR1 CTR2"   (GPU box, repo root)
set -e
W=${1:-convnt}; B=${2:-64}; CTRS=${3:-SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/kpmc_${W}
rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --pmc $CTRS -d $OUT/c -o run --output-format csv -- python3 $ROOT/tools/kprof.py $W $B > $OUT/c.log 2>&1
cd $ROOT
python3 - "$OUT" <<'PY'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(sys.argv[1] + "/c/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: "%.4g" % (sum(v) / len(v)) for c, v in d.items()})
PY
