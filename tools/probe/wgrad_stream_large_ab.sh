#!/bin/bash
# same-box A/B: the second (weight-gradient) stream for LARGE problems (cfg2 / cfg4 keep one stream by default: engine.WgradStream.wanted)
for wl in cfg2 cfg4; do for v in 0 1 0 1; do
UMR_WGRAD_STREAM=$v python bench.py --workload $wl --steps 8 --warmup 3 --no-cpu-baseline --no-alt 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$wl UMR_WGRAD_STREAM=$v', round(d['value'],1), round(d['ms_per_step'],2), 'mem GB', d.get('peak_memory_gb'))"
done; done
