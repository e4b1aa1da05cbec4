#!/bin/bash
# LDS-side counters of the attention kernels at the cfg2 shape:  bash tools/probe/attn_lds_pmc.sh
ROOT=$(pwd); export TMPDIR=/tmp; cd /tmp
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_LDS_LOAD"; do
  rm -rf /tmp/attn_lds
  ATTN_ONLY=1 rocprofv3 --kernel-trace --pmc $set -d /tmp/attn_lds -o run --output-format csv -- python3 $ROOT/tools/attn_bench.py > /tmp/attn_lds.log 2>&1 || tail -3 /tmp/attn_lds.log
  python3 - <<'PY'
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob("/tmp/attn_lds/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "attn" in n and "prep" not in n:
            acc[n[n.find("attn"):][:28]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    print(k, {c: f"{sum(v) / len(v):.4g}" for c, v in sorted(d.items())})
PY
done
