"""RCCL smoke on the one-GPU box: a world-size-1 "nccl" process group (RCCL on ROCm) runs the exact collective calls the
data-parallel step makes -- asynchronous bucketed all-reduce of slices of a flat CUDA buffer issued while compute kernels
are queued, wait before the optimizer, MAX all-reduce of the step time, barrier with device_ids.  (RCCL refuses two ranks
on one device, so world > 1 over RCCL first runs on the driver's multi-GPU node; world-2 semantics are covered over gloo.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import os, sys, socket
sys.path.insert(0, %r)
import torch, torch.distributed as dist
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
from unmore_amd import ops
from unmore_amd.parallel import BucketedAllReduce
n = 3_000_000
flat = torch.arange(n, dtype=torch.float32, device=dev)
ref = flat.clone()
comm = BucketedAllReduce(flat, [0, 1_000_000, 1_000_000, 2_500_000, n])
comm.enabled, comm.world = True, 1          # world 1: the reduction is the identity, the RCCL kernels still run
a = torch.randn(4096, 512, device=dev).bfloat16(); w = torch.randn(512, 512, device=dev).bfloat16()
for k in range(comm.num_buckets):
    ops.gemm_nt(a, w, None)                 # compute queued around the collectives, as in backward
    comm.ready(k)
scale = comm.finish()
ops.adam_step(flat.clone(), flat, torch.zeros_like(flat), torch.zeros_like(flat), 1, grad_scale=scale)
# the traced exchange (bench.py runs one untimed step with it when world > 1): RCCL's completion stamped by a side stream, a reduced
# CU budget for the persistent GEMMs beside the collectives (ops.set_cu_budget), results unchanged
big = torch.randn(8192, 768, device=dev).bfloat16(); wb = torch.randn(768, 768, device=dev).bfloat16()
c_full = ops.gemm_nt(big, wb, None)
ops.set_cu_budget(224)
comm.trace = True
for k in range(comm.num_buckets):
    c_b = ops.gemm_nt(big, wb, None)
    comm.ready(k)
comm.finish()
rep = comm.trace_report()
ops.set_cu_budget(0)
assert torch.equal(flat, ref) and torch.equal(c_full, c_b)
assert [r["bucket"] for r in rep["buckets"]] == [0, 2, 3] and rep["clock"] == "device events", rep
assert all(0.0 <= r["issue_ms"] <= r["done_ms"] for r in rep["buckets"]) and rep["exposed_ms"] >= 0.0, rep
print("TRACE", rep)
t = torch.tensor([1.25], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier(device_ids=[0])
torch.cuda.synchronize()
assert scale == 1.0 and torch.equal(flat, ref) and float(t.item()) == 1.25
print("RCCL_OK", ".".join(map(str, torch.cuda.nccl.version())))
dist.destroy_process_group()
''' % ROOT


def test_rccl_world1_runs_the_dp_collectives():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", SCRIPT], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])
    print(r.stdout.strip().splitlines()[-1])
