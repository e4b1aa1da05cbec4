#!/bin/bash
# every non-headline bench line in one go (GPU box): bash tools/probe/all_lines.sh > gpurun_out/lines.jsonl
set -e
for wl in cfg1 cfg4 ref; do python bench.py --workload $wl --no-cpu-baseline 2>/dev/null | grep '"metric"'; done
python bench.py --workload ref --dtype fp32 --no-cpu-baseline 2>/dev/null | grep '"metric"'
python bench.py --workload cfg5 2>/dev/null | grep '"metric"'
