#!/bin/bash
# same-box A/B: C stores of the persistent kernel's fast epilogue with the nt (streaming) bit.  GPU box, repo root.
bash tools/probe/build_exp_lib.sh gemm_nt256p.hip -DUMR_EXP_NT_STORE > /dev/null || exit 1
for i in 1 2; do
echo "== default"; timeout -k 10 200 python tools/probe/red_bench.py 2>&1 | grep -E "ms "
echo "== nt stores"; UMR_LIB=unmore_amd/lib/libumr_exp.so timeout -k 10 200 python tools/probe/red_bench.py 2>&1 | grep -E "ms "
done
echo "== default"; timeout -k 10 200 python tools/kbench.py 64 2>&1 | grep -E "conv3x3|1x1"
echo "== nt stores";  UMR_LIB=unmore_amd/lib/libumr_exp.so timeout -k 10 200 python tools/kbench.py 64 2>&1 | grep -E "conv3x3|1x1"
for L in unmore_amd/lib/libumr.so unmore_amd/lib/libumr_exp.so unmore_amd/lib/libumr.so unmore_amd/lib/libumr_exp.so; do UMR_LIB=$L python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-alt 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$L',round(d['value'],1),round(d['ms_per_step'],2),d['roofline']['frac'])"; done
