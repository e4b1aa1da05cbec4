"""world_size-2 gloo test (CPU) of the data-parallel gradient exchange used by TrainStep:
bucketed asynchronous all-reduce of a flat gradient buffer + 1/world scaling."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from unmore_amd.parallel import BucketedAllReduce
    n = 1000
    g = torch.Generator().manual_seed(100 + rank)
    flat = torch.randn(n, generator=g)
    local = flat.clone()
    bounds = [0, 64, 64, 640, 1000]  # includes an empty bucket
    comm = BucketedAllReduce(flat, bounds)
    assert comm.enabled and comm.world == world and comm.num_buckets == 4
    for k in range(comm.num_buckets):  # backward-completion order
        comm.ready(k)
    scale = comm.finish()
    # a second, traced exchange (what bench.py records for one untimed step when world > 1): same sums, plus per-bucket timestamps
    reduced = flat.clone()
    flat.copy_(local)
    comm.trace = True
    for k in range(comm.num_buckets):
        comm.ready(k)
    comm.finish()
    rep = comm.trace_report()
    assert torch.equal(flat, reduced)
    assert [r["bucket"] for r in rep["buckets"]] == [0, 2, 3] and rep["clock"] == "host clock"       # the empty bucket launches nothing
    assert all(0.0 <= r["issue_ms"] <= r["done_ms"] for r in rep["buckets"]) and rep["exposed_ms"] >= 0.0
    assert rep["buckets"][1]["mbytes"] == round(576 * 4 / 2 ** 20, 2) and comm.trace_report() is None   # cleared
    torch.save({"local": local, "reduced": reduced, "scale": scale}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_world2(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(os.path.join(tmp_path, f"r{i}.pt")) for i in range(world)]
    expect = r[0]["local"] + r[1]["local"]
    for i in range(world):
        torch.testing.assert_close(r[i]["reduced"], expect)
        assert r[i]["scale"] == 0.5
    # averaged gradient == gradient of the mean loss over the global batch
    torch.testing.assert_close(r[0]["reduced"] * r[0]["scale"], (r[0]["local"] + r[1]["local"]) / 2)


def test_single_process_is_a_noop():
    from unmore_amd.parallel import BucketedAllReduce
    flat = torch.arange(10.0)
    comm = BucketedAllReduce(flat, [0, 4, 10])
    assert not comm.enabled
    comm.ready(0)
    comm.ready(1)
    assert comm.finish() == 1.0
    assert torch.equal(flat, torch.arange(10.0))


def _worker_wire(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from unmore_amd.parallel import BucketedAllReduce
    n = 100_000
    g = torch.Generator().manual_seed(200 + rank)
    # gradient-like values over ten binades, the ranks' values correlated (same sign mostly, as gradients of two half-batches are)
    base = torch.randn(n, generator=torch.Generator().manual_seed(7)) * torch.exp2(torch.randint(-10, 1, (n,), generator=torch.Generator().manual_seed(8)).float())
    flat = base * (1.0 + 0.3 * torch.randn(n, generator=g))
    local = flat.clone()
    bounds = [0, 4096, 4096, 50_000, n]
    comm = BucketedAllReduce(flat, bounds, wire_dtype=torch.bfloat16)
    assert comm.enabled and comm.wire is not None and comm.wire.dtype == torch.bfloat16 and comm.grad_scale == 1.0
    comm.trace = True
    for k in range(comm.num_buckets):
        comm.ready(k)
    scale = comm.finish()
    rep = comm.trace_report()
    assert scale == 1.0 and rep["buckets"][0]["mbytes"] == round(4096 * 2 / 2 ** 20, 2)      # two bytes per element on the wire
    torch.save({"local": local, "reduced": flat.clone()}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_bf16_wire_exchange_against_the_f32_exchange_world2(tmp_path):
    """the bf16 gradient wire (BucketedAllReduce(wire_dtype=torch.bfloat16)): every rank ends with the same buffer, equal to the mean of
    the ranks' f32 gradients -- what the f32 exchange x 1/world gives -- to bf16 rounding: relative L2 <= 6e-3 over the buffer and per
    bucket, every element within 2^-7 of its value (three roundings of 2^-9 each), and the scale handed to the optimizer is 1"""
    world, port = 2, _free_port()
    mp.spawn(_worker_wire, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(os.path.join(tmp_path, f"r{i}.pt")) for i in range(world)]
    assert torch.equal(r[0]["reduced"], r[1]["reduced"])
    mean = ((r[0]["local"].double() + r[1]["local"].double()) / 2)
    got = r[0]["reduced"].double()
    rel = float((got - mean).norm() / mean.norm())
    assert rel <= 6e-3, rel
    for lo, hi in ((0, 4096), (4096, 50_000), (50_000, 100_000)):
        assert float((got[lo:hi] - mean[lo:hi]).norm() / mean[lo:hi].norm()) <= 6e-3
    tol = 2.0 ** -7 * torch.maximum(r[0]["local"].abs(), r[1]["local"].abs()).double() / 2 * 2
    assert bool(((got - mean).abs() <= tol + 1e-30).all())


def _worker_wait(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from unmore_amd.parallel import BucketedAllReduce
    n = 900
    bounds = [0, 300, 600, 900]
    out = {}
    for wire in (None, torch.bfloat16):
        g = torch.Generator().manual_seed(7 + rank)
        flat = torch.randn(n, generator=g)
        local = flat.clone()
        comm = BucketedAllReduce(flat, bounds, wire_dtype=wire)
        comm.ready(0)
        comm.ready(1)
        comm.wait(0)                      # bucket 0's exchange is complete (and, bf16 wire, widened) before the others are even sent
        first = flat[:300].clone()
        untouched = flat[600:].clone()
        comm.wait(0)                      # waiting twice, or for a bucket that was never sent, is a no-op
        comm.wait(2)
        comm.ready(2)
        scale = comm.finish()
        assert not comm._pending and not comm._sent
        out["f32" if wire is None else "bf16"] = dict(local=local, first=first, untouched=untouched, final=flat.clone(), scale=scale)
    torch.save(out, os.path.join(out_dir, f"w{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_waiting_for_one_bucket_before_finish_world2(tmp_path):
    """BucketedAllReduce.wait(k): what the stage-by-stage optimizer of small data-parallel steps calls (trainer.TrainStep) -- bucket k is
    exchanged (bf16 wire: and widened back into the gradient buffer) when it returns, the others are untouched, finish() covers the rest"""
    world, port = 2, _free_port()
    mp.spawn(_worker_wait, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(os.path.join(tmp_path, f"w{i}.pt")) for i in range(world)]
    total = r[0]["f32"]["local"] + r[1]["f32"]["local"]
    for i in range(world):
        a = r[i]["f32"]
        torch.testing.assert_close(a["first"], total[:300])
        assert torch.equal(a["untouched"], a["local"][600:]) and a["scale"] == 0.5
        torch.testing.assert_close(a["final"], total)
        b = r[i]["bf16"]
        mean = total / 2
        assert b["scale"] == 1.0 and torch.equal(b["untouched"], b["local"][600:])
        assert float((b["first"] - mean[:300]).norm() / mean[:300].norm()) <= 6e-3
        assert float((b["final"] - mean).norm() / mean.norm()) <= 6e-3
        assert torch.equal(b["final"][:300], b["first"])          # widened once, in wait(); finish() did not touch it again
