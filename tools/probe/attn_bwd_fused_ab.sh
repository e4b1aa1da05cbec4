#!/bin/bash
# same-box A/B: backward of short-sequence attention as one launch (default) against prep + dQ + dK/dV (UMR_ATTN_BWD_FUSED=0)
for i in 1 2 3; do
for v in 0 1; do
UMR_ATTN_BWD_FUSED=$v python bench.py --workload ref --steps 20 --warmup 5 --no-cpu-baseline --no-alt 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('UMR_ATTN_BWD_FUSED=$v', round(d['value'],1), 'images/s', round(d['ms_per_step'],2), 'ms/step')"
done; done
