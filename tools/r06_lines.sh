#!/bin/bash
# Round 6: every bench line on ONE box, plus the same-box A/B of the copy-writing optimizer launch on the reference recipe.
#   bash tools/r06_lines.sh      (GPU box, repo root; writes gpurun_out/profiles_out/r06_bench_lines.jsonl and r06_adam_pack_ab.txt)
mkdir -p gpurun_out/profiles_out
OUT=gpurun_out/profiles_out/r06_bench_lines.jsonl
: > $OUT
python bench.py --steps 20 --warmup 5 2> gpurun_out/r6_line_cfg2.err | grep '"metric"' >> $OUT; echo "cfg2 done"
python bench.py --workload cfg4 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | grep '"metric"' >> $OUT; echo "cfg4 done"
python bench.py --workload cfg1 --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '"metric"' >> $OUT; echo "cfg1 done"
python bench.py --workload ref --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '"metric"' >> $OUT; echo "ref done"
python bench.py --workload ref --graphs off --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '"metric"' >> $OUT; echo "ref eager done"
python bench.py --workload ref --dtype fp32 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '"metric"' >> $OUT; echo "ref fp32 done"
python bench.py --workload cfg5 --steps 3 --warmup 1 2> gpurun_out/r6_line_cfg5.err | grep '"metric"' >> $OUT; echo "cfg5 done"
AB=gpurun_out/profiles_out/r06_adam_pack_ab.txt
echo "# reference recipe (dpt_large, 128x128, batch 20, bf16, chain of per-stage graphs), same box, alternating: UMR_ADAM_PACK=0 (Adam launch + batched refresh of the" > $AB
echo "# packed copies per stage, round 5) against 1 (one optimizer launch per stage that writes the Linear weights' bf16 copies itself, round 6)" >> $AB
for i in 1 2 3; do for v in 0 1; do
UMR_ADAM_PACK=$v python bench.py --workload ref --steps 50 --warmup 5 --no-cpu-baseline --no-alt 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('UMR_ADAM_PACK=$v', round(d['value'],1), 'images/s', round(d['ms_per_step'],3), 'ms/step')" >> $AB
done; done
cat $AB
