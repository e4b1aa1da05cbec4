"""From a rocprofv3 --kernel-trace csv: the distribution of one kernel's durations and what ran right before its longest calls.
    python tools/probe/long_calls.py <dir with *_kernel_trace.csv> <kernel name substring>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
key = sys.argv[2]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), i) for i, r in enumerate(rows) if key in r["Kernel_Name"]]
tot = sum(x for x, _ in d)
print(len(d), "calls, total ms", tot / 1e6)
for lo, hi in ((0, 2e4), (2e4, 1e5), (1e5, 1e6), (1e6, 1e7), (1e7, 1e9)):
    sel = [x for x, _ in d if lo <= x < hi]
    print(f"  {lo/1e3:8.0f} .. {hi/1e3:8.0f} us: {len(sel):6d} calls, {sum(sel)/1e6:9.2f} ms")
for x, i in sorted(d, reverse=True)[:6]:
    r = rows[i]
    print(f"long call {x/1e3:9.1f} us  grid {r.get('Grid_Size_X', r.get('Grid_Size'))} wg {r.get('Workgroup_Size_X', r.get('Workgroup_Size'))} queue {r.get('Queue_Id')}")
    for j in range(max(0, i - 3), min(len(rows), i + 2)):
        q = rows[j]
        print(f"      {'>>' if j == i else '  '} start {(int(q['Start_Timestamp']) - int(r['Start_Timestamp']))/1e3:10.1f} us  dur {(int(q['End_Timestamp']) - int(q['Start_Timestamp']))/1e3:9.1f} us  q{q.get('Queue_Id')}  grid {q.get('Grid_Size_X', q.get('Grid_Size'))}  {q['Kernel_Name'][:80]}")
