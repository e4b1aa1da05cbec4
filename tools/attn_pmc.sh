#!/bin/bash
# SQ counters of the bf16 attention kernels at the cfg2 shape (GPU box, repo root) -- the evidence behind DESIGN.md section 5's
# "issue-bound at head dim 64": instruction counts per class and the busy / wait split of the wave cycles.
#   bash tools/attn_pmc.sh > gpurun_out/attn_pmc.txt
set -e
ROOT=$(pwd); OUT=$ROOT/gpurun_out/attn_pmc
rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $OUT/a -o run --output-format csv -- python3 $ROOT/tools/attn_bench.py > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $OUT/b -o run --output-format csv -- python3 $ROOT/tools/attn_bench.py > $OUT/b.log 2>&1
rocprofv3 --kernel-trace --pmc MfmaUtil -d $OUT/c -o run --output-format csv -- python3 $ROOT/tools/attn_bench.py > $OUT/c.log 2>&1
cd $ROOT
python3 - "$OUT" <<'PY'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(sys.argv[1] + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "attn" in n:
            k = n[n.find("attn"):][:34]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    m = {c: sum(v) / len(v) for c, v in d.items()}
    print(k)
    for c in sorted(m):
        print(f"    {c:28s} {m[c]:14.4g}")
    if "SQ_WAVE_CYCLES" in m and "SQ_WAVES" in m:
        w = m["SQ_WAVES"]
        print(f"    per wave: VALU {m.get('SQ_INSTS_VALU', 0) / w:8.1f}  MFMA {m.get('SQ_INSTS_MFMA', 0) / w:8.1f}  LDS {m.get('SQ_INSTS_LDS', 0) / w:8.1f}  SALU {m.get('SQ_INSTS_SALU', 0) / w:8.1f}"
              f"  | wave cycles (x4) {4 * m['SQ_WAVE_CYCLES'] / w:10.0f}  of which waiting {100 * m.get('SQ_WAIT_ANY', 0) / m['SQ_WAVE_CYCLES']:5.1f} %  issue-stalled {100 * m.get('SQ_WAIT_INST_ANY', 0) / m['SQ_WAVE_CYCLES']:5.1f} %  issuing {100 * m.get('SQ_ACTIVE_INST_ANY', 0) / m['SQ_WAVE_CYCLES']:5.1f} %")
PY
