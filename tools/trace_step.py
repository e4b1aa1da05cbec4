"""Condense a rocprofv3 --kernel-trace CSV of bench.py into the anatomy of ONE steady-state step: per stream busy time and gaps, the
kernels by total time with their grids, and (with --list) every launch in order.  usage: python tools/trace_step.py <kernel_trace.csv>
[--step-marker adam_set_hyper] [--list] [--top 30]"""
import argparse, csv, collections, re


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", n)
    n = n.replace("void ", "")
    return n[:100]


ap = argparse.ArgumentParser()
ap.add_argument("csv")
ap.add_argument("--step-marker", default="adam_set_hyper")
ap.add_argument("--list", action="store_true")
ap.add_argument("--top", type=int, default=30)
a = ap.parse_args()
rows = list(csv.DictReader(open(a.csv)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if a.step_marker in r["Kernel_Name"]]
assert len(marks) >= 2, "need at least two step markers"
lo, hi = marks[-2], marks[-1]
step = rows[lo:hi]
t0, t1 = int(step[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in step)
print(f"step: {len(step)} launches, wall {1e-6 * (t1 - t0):.3f} ms (marker to marker start: {1e-6 * (int(rows[hi]['Start_Timestamp']) - t0):.3f} ms)")
by_q = collections.defaultdict(list)
for r in step:
    by_q[r.get("Queue_Id", "0")].append(r)
for q, rs in by_q.items():
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs)
    gaps = [int(b["Start_Timestamp"]) - int(a_["End_Timestamp"]) for a_, b in zip(rs, rs[1:])]
    pos = [g for g in gaps if g > 0]
    print(f"queue {q}: {len(rs)} launches, busy {1e-6 * busy:.3f} ms, span {1e-6 * (int(rs[-1]['End_Timestamp']) - int(rs[0]['Start_Timestamp'])):.3f} ms, "
          f"gaps: sum {1e-6 * sum(pos):.3f} ms, median {sorted(pos)[len(pos) // 2] / 1e3 if pos else 0:.1f} us")
# union of busy intervals (any queue)
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in step)
cov, cur_s, cur_e = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > cur_e:
        cov += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
cov += cur_e - cur_s
print(f"GPU busy (union over queues) {1e-6 * cov:.3f} ms = {100.0 * cov / (t1 - t0):.1f} % of the step; sum of kernel durations {1e-6 * sum(e - s for s, e in iv):.3f} ms")
agg = collections.defaultdict(lambda: [0, 0, set()])
for r in step:
    k = short(r["Kernel_Name"])
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    agg[k][0] += 1
    agg[k][1] += d
    agg[k][2].add(int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])))
print(f"{'kernel':100s} {'n':>5s} {'ms':>8s} {'avg us':>8s}  workgroups")
for k, (n, d, g) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:a.top]:
    gs = sorted(g)
    print(f"{k:100s} {n:5d} {1e-6 * d:8.3f} {1e-3 * d / n:8.1f}  {gs[:6]}{'...' if len(gs) > 6 else ''}")
if a.list:
    prev_end = {}
    for r in step:
        q = r.get("Queue_Id", "0")
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - prev_end[q]) / 1e3 if q in prev_end else 0.0
        prev_end[q] = e
        print(f"{1e-3 * (s - t0):10.1f} us q{q} {1e-3 * (e - s):8.1f} us gap {gap:7.1f}  wg {int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])):6d} x{r.get('Grid_Size_Y', '1')}  {short(r['Kernel_Name'])[:80]}")
