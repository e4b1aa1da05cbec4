"""bench.py end to end on the one-GPU box: the plain `--gpus 2` call (self-spawned ranks, gloo transport, both ranks on the
one device, miniature workload) prints ONE line with n_gpus 2, and the single-GPU lines of the small workloads carry the
contract's keys.  RCCL itself needs one device per rank and is first exercised by the driver's multi-GPU run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(*args, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, BENCH, *args], env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_two_ranks_spawned_by_bench_itself():
    res = _run("--gpus", "2", "--workload", "tiny", "--backend", "gloo", "--steps", "3", "--warmup", "1")
    assert res["n_gpus"] == 2 and res["config"]["global_batch"] == 4 and res["config"]["parallelism"] == "dp2"
    assert res["collective"]["backend"] == "gloo" and res["collective"]["world"] == 2
    assert res["value"] > 0 and res["scaling"] == "weak" and "cpu_baseline" not in res


def test_two_ranks_with_the_bf16_gradient_wire():
    """`--dp-wire bf16` end to end (2 gloo ranks on the one GPU, miniature workload): the line says which wire was used, the step runs and
    trains (finite loss); the exchange's accuracy is tests/test_parallel_cpu.py's and tests/test_dp_gpu.py's business"""
    res = _run("--gpus", "2", "--workload", "tiny", "--backend", "gloo", "--steps", "3", "--warmup", "1", "--dp-wire", "bf16")
    assert res["n_gpus"] == 2 and res["collective"]["gradient_wire"] == "bf16" and res["value"] > 0
    assert res["final_loss"] == res["final_loss"] and 0 < res["final_loss"] < 100
    tr = res.get("allreduce_trace_rank0")
    assert tr and "buckets" in tr and tr["buckets"][0]["mbytes"] > 0


@pytest.mark.parametrize("workload,extra", [("tiny", []), ("cfg1", []), ("cfg5", ["--steps", "1", "--warmup", "1"]),
                                            ("cfg4", ["--steps", "1", "--warmup", "1", "--no-alt"])])
def test_single_gpu_lines_carry_the_contract(workload, extra):
    res = _run("--workload", workload, "--no-cpu-baseline", *(extra or ["--steps", "2", "--warmup", "1"]))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in res, k
    assert res["n_gpus"] == 1 and res["value"] > 0 and "workload" in res["config"]
    rf = res["roofline"]
    assert rf["bound"] == "mfma" and rf["launches_timed"] > 0 and 0 < rf["frac"] < 1
    if workload == "cfg4":
        # BASELINE configs[3] at workload size: ViT-L/14 518x518 batch 16 (1370 tokens per image, 4.3 M pixels)
        assert res["config"]["per_gpu_batch"] == 16 and res["config"]["image"] == [518, 518] and res["dtype"] == "bf16"
        assert res["final_loss"] == res["final_loss"] and 0 < res["final_loss"] < 100      # finite
    if workload == "cfg5":
        assert res["config"]["proposals_per_image"] == 1225 and res["crops_per_sec"] > 0
        # BASELINE configs[4] ("peak-picking bit-exact vs CPU"): the headline is the fp32 parity mode on a net whose score maps are
        # NOT empty (at least half of the proposals have a peak), bf16 is reported beside it
        assert res["dtype"] == "fp32" and res["maps_with_peak"] >= 1225 // 2
        assert res["alt_bf16"]["value"] > res["value"]
        # round 6: the headline is the certificate-driven sweep; it returns the full six-term sweep's peak index for EVERY proposal
        sp, full = res["sweep_precision"], res["alt_fp32_full_precision"]
        assert sp["mode"] == "certified" and 0.0 <= sp["rerun_fraction"] < 0.5, sp
        assert full["peak_index_differs_from_headline"] == 0 and full["max_abs_amax_difference_vs_headline"] <= 1.5e-4, full
