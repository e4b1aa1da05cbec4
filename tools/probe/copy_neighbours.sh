#!/bin/bash
# which kernels run right before / after the runtime's __amd_rocclr_copyBuffer launches of a train step?
#   bash tools/probe/copy_neighbours.sh   -> gpurun_out/copy_neighbours.txt
ROOT=$(pwd)
export TMPDIR=/tmp
rm -rf gpurun_out/prof_cn
cd /tmp
rocprofv3 --kernel-trace -d $ROOT/gpurun_out/prof_cn -o run --output-format csv -- python3 $ROOT/bench.py --workload ref --steps 2 --warmup 1 --no-cpu-baseline --no-alt > $ROOT/gpurun_out/prof_cn.log 2>&1
cd $ROOT
python3 - <<'PY' > gpurun_out/copy_neighbours.txt
import csv, glob, collections
f = glob.glob("gpurun_out/prof_cn/**/*kernel_trace.csv", recursive=True)[0]
allrows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
print("columns:", list(allrows[0].keys()))
# the last step only (after the last loss kernel but one), per queue: launches on one queue run in order
qk = "Queue_Id" if "Queue_Id" in allrows[0] else "Stream_Id"
loss = [i for i, r in enumerate(allrows) if "loss_kernel" in r["Kernel_Name"]]
allrows = allrows[loss[-2]:loss[-1]] if len(loss) >= 2 else allrows
print("launches in the step:", len(allrows), " copyBuffer:", sum("copyBuffer" in r["Kernel_Name"] for r in allrows))
for q in sorted(set(r[qk] for r in allrows)):
    rows = [r for r in allrows if r[qk] == q]
    names = [r["Kernel_Name"][:60] for r in rows]
    pairs = collections.Counter()
    for i, n in enumerate(names):
        if "copyBuffer" in n:
            pairs[(names[i - 1] if i else "-", names[i + 1] if i + 1 < len(names) else "-", rows[i].get("Grid_Size_X", rows[i].get("Grid_Size", "?")))] += 1
    print("queue", q, "launches", len(rows))
    for (a, b, g), c in pairs.most_common(12):
        print(f"{c:5d}  grid {g:>8s}  before: {a:60s}  after: {b}")
PY
rm -rf gpurun_out/prof_cn
