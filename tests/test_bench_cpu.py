"""bench.py's N > 1 launch path on CPU (gloo, world_size 2): called plainly (`python bench.py --gpus 2`) the parent starts the
rank processes itself; called the way the driver does it (torch.distributed.run) it reads RANK / WORLD_SIZE from the
environment.  `--rehearse` = the plumbing around the model only (rendezvous, barrier, bucketed all-reduce, max-over-ranks
timing); the model itself needs the MI355X (tests/test_bench_gpu.py)."""
import json
import os
import time
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "2"
    return env


def _one_json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_bench_spawns_its_own_ranks_when_called_plainly():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--rehearse", "--workload", "tiny", "--steps", "3",
                        "--warmup", "1"], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    res = _one_json_line(r.stdout)
    assert res["n_gpus"] == 2 and res["steps"] == 3 and res["rehearsal"] is True
    assert res["collective"] == {"backend": "gloo", "world": 2, "gradient_wire": "f32"}
    assert res["allreduce_correct"] is True and res["ms_per_step"] > 0


def test_bench_under_torch_distributed_run():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), BENCH, "--gpus", "2", "--backend", "gloo", "--rehearse", "--workload", "tiny",
           "--steps", "2", "--warmup", "1"]
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    res = _one_json_line(r.stdout)
    assert res["n_gpus"] == 2 and res["allreduce_correct"] is True


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    env = dict(_env(), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--rehearse", "--backend", "gloo", "--workload", "tiny"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


def test_a_dying_rank_ends_the_plain_launch_within_seconds():
    """rank 1 exits before the rendezvous; rank 0 would wait in it for gloo's 30-minute timeout.  The parent must notice the
    exit, terminate rank 0, print rank 1's stderr and return non-zero quickly, without a JSON line on stdout."""
    t0 = time.time()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--rehearse", "--workload", "tiny", "--steps", "2",
                        "--warmup", "1", "--fail-rank", "1"], env=_env(), capture_output=True, text=True, timeout=120)
    dt = time.time() - t0
    assert r.returncode == 3, (r.returncode, r.stderr[-2000:])
    assert dt < 30, dt
    assert "rank 1 exited with status 3" in r.stderr and "--fail-rank requested" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_a_dying_rank_zero_ends_the_plain_launch_too():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--rehearse", "--workload", "tiny", "--steps", "2",
                        "--warmup", "1", "--fail-rank", "0"], env=_env(), capture_output=True, text=True, timeout=120)
    assert r.returncode == 3 and "rank 0 exited with status 3" in r.stderr


def test_a_hung_launch_hits_the_overall_deadline():
    """a port nobody listens on for rank 1 is simulated by a tiny deadline: the parent kills what it started and exits 124"""
    t0 = time.time()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--rehearse", "--workload", "cfg2", "--steps", "2000",
                        "--warmup", "1", "--launch-timeout", "3"], env=_env(), capture_output=True, text=True, timeout=120)
    assert r.returncode == 124, (r.returncode, r.stderr[-2000:])
    assert time.time() - t0 < 40 and "deadline of 3 s passed" in r.stderr


def test_eight_rank_rehearsal_exchanges_the_full_gradient_buffer():
    """The first 8-GPU run's plumbing, on CPU: 8 ranks (gloo), the ViT-B workload's whole flat gradient buffer (115.4 M elements)
    in its 16 backward-completion buckets, correct sums on every element, ONE line with 8 per-rank entries."""
    env = dict(_env(), OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--backend", "gloo", "--rehearse", "--workload", "cfg2", "--steps", "2",
                        "--warmup", "1"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    res = _one_json_line(r.stdout)
    assert res["n_gpus"] == 8 and res["collective"] == {"backend": "gloo", "world": 8, "gradient_wire": "f32"}
    assert res["allreduce_correct"] is True and res["allreduce_buckets"] == 16
    assert 115_000_000 < res["allreduce_elements"] < 116_000_000
    assert len(res["per_rank_ms_per_step"]) == 8 and all(t > 0 for t in res["per_rank_ms_per_step"])


def test_a_dying_rank_five_of_eight_ends_the_launch_within_seconds():
    t0 = time.time()
    env = dict(_env(), OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--backend", "gloo", "--rehearse", "--workload", "tiny", "--steps", "2",
                        "--warmup", "1", "--fail-rank", "5"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 3, (r.returncode, r.stderr[-2000:])
    assert time.time() - t0 < 30
    assert "rank 5 exited with status 3" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]
