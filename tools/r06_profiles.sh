#!/bin/bash
# Round 6 evidence run (GPU box, repo root): rocprofv3 kernel stats + PMC passes per workload, the attention ceiling probe, the head-mode soaks.
set -x
# (the raw rocprofv3 directories are deleted after each summary: gpurun copies back at most 64 MiB)
bash tools/profile_round.sh r06 cfg2 > gpurun_out/r6_prof_cfg2.log 2>&1; tail -1 gpurun_out/r6_prof_cfg2.log | cut -c1-200; rm -rf gpurun_out/prof_r06_cfg2
bash tools/profile_round.sh r06 cfg4 stats-only > gpurun_out/r6_prof_cfg4.log 2>&1; rm -rf gpurun_out/prof_r06_cfg4
bash tools/profile_round.sh r06 ref stats-only > gpurun_out/r6_prof_ref.log 2>&1; rm -rf gpurun_out/prof_r06_ref
bash tools/profile_round.sh r06 cfg5 > gpurun_out/r6_prof_cfg5.log 2>&1; tail -1 gpurun_out/r6_prof_cfg5.log | cut -c1-200; rm -rf gpurun_out/prof_r06_cfg5
python tools/probe/attn_ceiling.py > gpurun_out/profiles_out/r06_attention_ceiling.jsonl 2> gpurun_out/r6_attn_ceiling.err
python tools/head_mode_soak.py ref 2000 > gpurun_out/profiles_out/r06_head_mode_soak_ref_recipe_2000steps.txt 2>&1; tail -3 gpurun_out/profiles_out/r06_head_mode_soak_ref_recipe_2000steps.txt | cut -c1-300
python tools/head_mode_soak.py cfg2 800 > gpurun_out/profiles_out/r06_head_mode_soak_cfg2_800steps.txt 2>&1; tail -3 gpurun_out/profiles_out/r06_head_mode_soak_cfg2_800steps.txt | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | cut -c1-400
du -sh gpurun_out
