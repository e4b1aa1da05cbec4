#!/bin/bash
# Every bench line of the round on ONE box -> gpurun_out/profiles_out/<tag>_bench_lines.jsonl (copy to profiles/):
#   bash tools/round_lines.sh r05
TAG=${1:-r05}
mkdir -p gpurun_out/profiles_out
OUT=gpurun_out/profiles_out/${TAG}_bench_lines.jsonl
: > $OUT
run() { echo "== $*" >&2; python bench.py "$@" 2>gpurun_out/round_lines.err | grep '^{' >> $OUT; }
run                                             # cfg2: the driver's default line (with cpu_baseline and the alt legs)
run --workload cfg1
run --workload cfg4 --steps 5
run --workload ref --steps 10
run --workload ref --dtype fp32 --steps 10
run --workload ref --steps 10 --graphs off --no-cpu-baseline   # the eager two-stream schedule (A/B of the default: chain of per-stage graphs)
run --workload cfg5 --steps 2 --warmup 1
python - <<PY
import json
for l in open("$OUT"):
    r = json.loads(l)
    print(r["config"]["workload"][:70], "|", r["dtype"], "|", round(r["value"], 2), r["unit"], "| ms/step", round(r["ms_per_step"], 2),
          "| roofline", round(r["roofline"]["frac"], 3), "| cpu", (r.get("cpu_baseline") or {}).get("value"), "| graph", (r.get("hip_graph") or {}).get("replayed"))
PY
