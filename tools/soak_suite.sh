#!/bin/bash
# The soaks that found (or would have found) what no fixture does, as ONE command with a pass / fail answer (DESIGN.md section 8,
# "next" item 2): thousand-step runs of the reference recipe's shape in both head modes (the two attention bugs of round 6 showed at
# steps 921 ... 5750), the chain-of-graphs step against the eager step, the data-parallel step through RCCL at world 1 against the step
# without an exchange.  GPU box, repo root:   bash tools/soak_suite.sh [steps=4000]      (~4 min at the default)
STEPS=${1:-4000}
fail=0
run() {   # name, pass pattern, command...
  local name=$1 pat=$2; shift 2
  local log=gpurun_out/soak_$name.log
  mkdir -p gpurun_out
  timeout -k 10 900 "$@" > $log 2>&1
  local rc=$?
  if [ $rc -eq 0 ] && grep -q "$pat" $log; then echo "PASS $name: $(grep "$pat" $log | tail -1 | cut -c1-160)"; else echo "FAIL $name (exit $rc): $(tail -2 $log | cut -c1-200)"; fail=1; fi
}
run nan_hunt_collapsed "no non-finite loss or gradient" python tools/probe/nan_hunt_ref.py auto $STEPS
run nan_hunt_factored  "no non-finite loss or gradient" python tools/probe/nan_hunt_ref.py factored $STEPS
run chain_vs_eager     "differences: 0"                 python tools/probe/staged_soak.py $((STEPS / 10)) bf16
run rccl_world1_f32    "differences: 0"                 python tools/probe/dp_force_soak.py $((STEPS / 4)) f32
run rccl_world1_bf16   "differences: 0"                 python tools/probe/dp_force_soak.py $((STEPS / 10)) bf16 auto 100 1
exit $fail
