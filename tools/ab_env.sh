run() { python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-alt 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],2), round(d['roofline']['avg_launch_ms'],2))"; }
run base
UMR_NT256_WG_PER_CU=1 run wg1
UMR_NT256_WG_PER_CU=2 run wg2
UMR_NT256_WG_PER_CU=4 run wg4
run base
UMR_TN256_ROUNDS=2 run tnr2
UMR_TN256_ROUNDS=3 run tnr3
UMR_TN256_ROUNDS=6 run tnr6
run base
