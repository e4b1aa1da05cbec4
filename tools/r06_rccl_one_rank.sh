#!/bin/bash
# Round 6: the data-parallel path through the REAL backend on one GPU (a process group of one rank on "nccl" = RCCL; UMR_DP_FORCE=1), beside
# the same line without any exchange, same box:  bash tools/r06_rccl_one_rank.sh   -> gpurun_out/profiles_out/r06_rccl_one_rank.jsonl
mkdir -p gpurun_out/profiles_out
OUT=gpurun_out/profiles_out/r06_rccl_one_rank.jsonl
: > $OUT
for wl in cfg2 ref; do
  if [ $wl = cfg2 ]; then S="--steps 10 --warmup 3"; else S="--steps 50 --warmup 5"; fi
  python bench.py --workload $wl $S --no-cpu-baseline --no-alt 2> gpurun_out/rccl1_${wl}_plain.err | grep '"metric"' >> $OUT; echo "$wl plain done"
  UMR_DP_FORCE=1 python bench.py --workload $wl $S --no-cpu-baseline --no-alt 2> gpurun_out/rccl1_${wl}_f32.err | grep '"metric"' >> $OUT; echo "$wl rccl f32 done"
  UMR_DP_FORCE=1 python bench.py --workload $wl $S --dp-wire bf16 --no-cpu-baseline --no-alt 2> gpurun_out/rccl1_${wl}_bf16.err | grep '"metric"' >> $OUT; echo "$wl rccl bf16 done"
done
python - <<'P'
import json
for l in open("gpurun_out/profiles_out/r06_rccl_one_rank.jsonl"):
    d = json.loads(l)
    c = d.get("collective", {})
    t = d.get("allreduce_trace_rank0") or {}
    print(d["config"]["workload"], c.get("backend"), c.get("gradient_wire"), round(d["value"], 1), d["unit"], round(d["ms_per_step"], 2), "ms",
          "graphs:", d.get("graphs", {}).get("replayed") if isinstance(d.get("graphs"), dict) else d.get("graphs"),
          "buckets:", len(t.get("buckets", [])), "exposed_ms:", t.get("exposed_ms"), t.get("error"))
P
