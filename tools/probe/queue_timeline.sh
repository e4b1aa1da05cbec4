#!/bin/bash
# One train step of the reference recipe on the kernel trace: per hardware queue, how much of the step is kernels and how much is
# the gap between one kernel's end and the next one's start.   bash tools/probe/queue_timeline.sh [bench.py args]  -> gpurun_out/queue_timeline.txt
ROOT=$(pwd)
export TMPDIR=/tmp
rm -rf gpurun_out/prof_qt
ARGS=${@:---workload ref --steps 3 --warmup 2 --no-cpu-baseline --no-alt}
cd /tmp
rocprofv3 --kernel-trace -d $ROOT/gpurun_out/prof_qt -o run --output-format csv -- python3 $ROOT/bench.py $ARGS > $ROOT/gpurun_out/prof_qt.log 2>&1
cd $ROOT
python3 - <<'PY' > gpurun_out/queue_timeline.txt
import csv, glob, collections
f = glob.glob("gpurun_out/prof_qt/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
loss = [i for i, r in enumerate(rows) if "loss_kernel" in r["Kernel_Name"]]
rows = rows[loss[-2]:loss[-1]]
t0, t1 = int(rows[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rows)
print(f"one step (loss kernel to loss kernel): {(t1 - t0) / 1e6:.3f} ms, {len(rows)} launches")
for q in sorted(set(r["Queue_Id"] for r in rows)):
    rs = [r for r in rows if r["Queue_Id"] == q]
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs)
    gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(rs, rs[1:])]
    pos = [g for g in gaps if g > 0]
    small = [g for g in pos if g < 30000]
    print(f"queue {q}: {len(rs)} launches, kernels {busy / 1e6:.3f} ms, span {(int(rs[-1]['End_Timestamp']) - int(rs[0]['Start_Timestamp'])) / 1e6:.3f} ms, "
          f"gaps {sum(pos) / 1e6:.3f} ms in {len(pos)} (under 30 us: {sum(small) / 1e6:.3f} ms in {len(small)}, median {sorted(small)[len(small) // 2] / 1e3 if small else 0:.2f} us)")
    by = collections.Counter()
    for a, b, g in zip(rs, rs[1:], gaps):
        if 0 < g < 30000:
            by[a["Kernel_Name"][:48]] += g
    for k, v in by.most_common(8):
        print(f"      gap after {k:48s} {v / 1e6:.3f} ms")
    big = collections.Counter()
    bign = collections.Counter()
    for a, b, g in zip(rs, rs[1:], gaps):
        if g >= 30000:
            key = (a["Kernel_Name"][:44], b["Kernel_Name"][:44])
            big[key] += g
            bign[key] += 1
    for (ka, kb), v in big.most_common(14):
        print(f"      LONG gaps {bign[(ka, kb)]:3d} x, {v / 1e6:.3f} ms: after {ka:44s} before {kb}")
# overlap of the two busiest queues
PY
rm -rf gpurun_out/prof_qt
cat gpurun_out/queue_timeline.txt
