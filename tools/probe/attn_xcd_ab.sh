#!/bin/bash
# same-box A/B of the attention kernels' block map (UMR_ATTN_XCD=0: blockIdx order, 1: XCD-aware), alternating, 3 rounds, three shapes,
# then one FETCH_SIZE pass per setting at the cfg2 shape.   bash tools/probe/attn_xcd_ab.sh
ROOT=$(pwd)
for shape in 64,577,12 16,1370,16 20,65,16; do
  echo "== shape (B,N,heads) = $shape"
  for round in 1 2 3; do
    for x in 0 1; do
      printf "UMR_ATTN_XCD=%s  " $x
      UMR_ATTN_XCD=$x ATTN_SHAPE=$shape ATTN_ONLY=1 python tools/attn_bench.py 2>/dev/null | tr '\n' ' '
      echo
    done
  done
done
export TMPDIR=/tmp
for x in 0 1; do
  rm -rf /tmp/attn_pmc_$x
  (cd /tmp && UMR_ATTN_XCD=$x ATTN_ONLY=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/attn_pmc_$x -o run --output-format csv -- python3 $ROOT/tools/attn_bench.py > /dev/null 2>&1)
  python3 - $x <<'PY'
import csv, glob, sys, collections
x = sys.argv[1]
acc = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(f"/tmp/attn_pmc_{x}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") == "FETCH_SIZE" and "attn" in r["Kernel_Name"]:
            k = r["Kernel_Name"].split("(")[0][-40:]
            acc[k][0] += 1
            acc[k][1] += float(r["Counter_Value"])
for k, (n, v) in sorted(acc.items()):
    print(f"UMR_ATTN_XCD={x}  {k:42s} launches {n:3d}  fetched per launch {2.0 * 1024.0 * v / n / 1e9:6.3f} GB (FETCH_SIZE x 2)")
PY
done
