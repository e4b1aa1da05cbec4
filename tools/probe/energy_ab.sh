#!/bin/bash
# same-box energy table (GPU box, repo root): the default library and experimental builds / run-time switches, one process each
#   bash tools/probe/energy_ab.sh > gpurun_out/energy.jsonl
set -e
bash tools/probe/build_exp_lib.sh gemm_nt256p.hip -DUMR_EXP_NT_STORE > /dev/null
for k in conv_nt conv_nt_masked conv_tn g1x1_nt conv_x3; do
  timeout -k 10 120 python tools/energy_probe.py --kernel $k --tag default
done
UMR_LIB=unmore_amd/lib/libumr_exp.so timeout -k 10 120 python tools/energy_probe.py --kernel conv_nt --tag nt_store_C
UMR_LIB=unmore_amd/lib/libumr_exp.so timeout -k 10 120 python tools/energy_probe.py --kernel g1x1_nt --tag nt_store_C
UMR_NT256_STAGGER=0 timeout -k 10 120 python tools/energy_probe.py --kernel conv_nt --tag no_stagger
UMR_NT256_WG_PER_CU=1 timeout -k 10 120 python tools/energy_probe.py --kernel conv_nt --tag one_wg_per_cu
timeout -k 10 120 python tools/energy_probe.py --kernel conv_nt --tag default_again
