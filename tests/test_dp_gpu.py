"""Data-parallel TrainStep, 2 ranks on the one visible GPU (gloo transport, CUDA tensors): after one step the
weights equal those of a single process that saw the concatenated batch (gradients averaged over ranks ==
gradient of the mean loss over the global batch)."""
import os
import socket
from argparse import Namespace

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ARGS = Namespace(use_bg_sdf=True, sdf_activation="tanh")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_net(backbone="dpt_tiny", tag="tiny"):
    from unmore_amd.hashrng import hash_init
    from unmore_amd.objectness_net import ObjectnessNet
    net = ObjectnessNet("cuda:0", 64, backbone, ARGS)
    sd = {k: torch.from_numpy(hash_init(k, tuple(v.shape), tag)) for k, v in net.state_dict().items()}
    net.load_state_dict(sd, strict=True)
    return net.to("cuda:0")


def _worker(rank, world, port, out_dir, backbone, tag, H, W):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from unmore_amd import synth
    from unmore_amd.trainer import TrainStep
    net = _make_net(backbone, tag)
    img, cf, sdf, sal = (torch.from_numpy(a).cuda() for a in synth.make_batch(4, H, W, seed=7))
    sl = slice(rank * 2, rank * 2 + 2)
    step = TrainStep(net, lr=1e-3)
    assert step.comm.enabled and step.comm.world == 2
    step.step(img[sl], cf[sl], sdf[sl], sal[sl])
    torch.cuda.synchronize()
    # every rank holds the same exchanged gradient and the same updated weights, bit for bit
    for buf in (step.flat_g, step.flat_p):
        ref = buf.clone()
        dist.broadcast(ref, 0)
        assert torch.equal(ref, buf), "ranks diverged"
    if rank == 0:
        torch.save({"flat_g": step.flat_g.cpu(), "flat_p": step.flat_p.cpu(), "buckets": len(step.comm.bounds) - 1}, os.path.join(out_dir, "r0.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("backbone,tag,H,W", [("dpt_tiny", "tiny", 64, 64), ("dpt_base", "base", 64, 96)])
def test_two_rank_step_equals_single_process_on_global_batch(tmp_path, backbone, tag, H, W):
    """The exchanged flat gradient itself (TrainStep.flat_g after the bucketed all-reduce: the SUM over ranks; 1/world is folded
    into Adam) against the gradient a single process computes on the concatenated batch -- per bucket and per parameter tensor --
    on dpt_tiny (7 buckets) and on the benchmark's ViT-B wiring (dpt_base: the step's 16 real buckets, 115.4 M elements).
    Every image's forward is the same arithmetic in either split; the weight gradients sum the same per-pixel terms in another
    order (2 + 2 images vs 4): f32 rounding, 1e-6-ish."""
    from unmore_amd import synth
    from unmore_amd.trainer import TrainStep, flat_layout
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), backbone, tag, H, W), nprocs=world, join=True)
    r0 = torch.load(os.path.join(tmp_path, "r0.pt"))
    net = _make_net(backbone, tag)
    img, cf, sdf, sal = (torch.from_numpy(a).cuda() for a in synth.make_batch(4, H, W, seed=7))
    step = TrainStep(net, lr=1e-3)
    step.step(img, cf, sdf, sal)
    torch.cuda.synchronize()
    offs, bounds, stage_bucket = flat_layout(net)
    assert r0["buckets"] == len(bounds) - 1 == (16 if backbone == "dpt_base" else len(bounds) - 1)
    g_dp = r0["flat_g"].double() / world           # mean over ranks == gradient of the mean loss over the global batch
    g_1 = step.flat_g.cpu().double()
    assert g_dp.shape == g_1.shape and float(g_1.abs().max()) > 0
    worst = (0.0, "")
    for k in range(len(bounds) - 1):
        a, b = g_dp[bounds[k]:bounds[k + 1]], g_1[bounds[k]:bounds[k + 1]]
        e = float((a - b).norm() / (b.norm() + 1e-300))
        assert e <= 5e-5, (f"bucket {k}", e)
    named = dict(net.named_parameters())
    for n, o in offs.items():
        cnt = named[n].numel()
        a, b = g_dp[o:o + cnt], g_1[o:o + cnt]
        if cnt >= 64 and float(b.norm()) > 0:
            e = float((a - b).norm() / b.norm())
            worst = max(worst, (e, n))
            assert e <= 2e-4, (n, e)
    e_inf = float((g_dp - g_1).abs().max() / g_1.abs().max())
    print(f"{backbone}: {len(bounds) - 1} buckets, {g_1.numel()} elements; worst per-tensor rel L2 {worst[0]:.2e} ({worst[1]}); max-norm error {e_inf:.2e} of max|g|")
    assert e_inf <= 1e-3
    # and the update that follows from it: Adam's first step moves every weight by ~lr * sign(g); only weights whose gradient is
    # ~0 may land on the other side
    p_dp, p_1 = r0["flat_p"], step.flat_p.cpu()
    bad = int(((p_dp - p_1).abs() > 2e-4).sum())
    assert bad <= 2e-3 * p_1.numel(), (bad, p_1.numel())


def _worker_chain(rank, world, port, out_dir, backbone, tag, B, H, W, dtype_name):
    """the three variants one after the other in ONE process group (a process start + import per variant costs more than its steps)"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from argparse import Namespace
    from unmore_amd import graphs, synth
    from unmore_amd.objectness_net import ObjectnessNet
    from unmore_amd.trainer import TrainStep
    torch.manual_seed(0)                         # identical initial weights on both ranks and in every variant (default nn init)
    init = {k: v.clone() for k, v in ObjectnessNet("cpu", H, backbone, Namespace(use_bg_sdf=True, sdf_activation="tanh")).state_dict().items()}
    keep = None
    for mode, wire in (("off", "f32"), ("auto", "f32"), ("auto", "bf16")):
        net = ObjectnessNet("cuda:0", H, backbone, Namespace(use_bg_sdf=True, sdf_activation="tanh"))
        net.load_state_dict(init, strict=True)
        net = net.to("cuda:0")
        net.set_compute_dtype(getattr(torch, dtype_name))
        step = TrainStep(net, lr=1e-4, grad_wire_dtype=(torch.bfloat16 if wire == "bf16" else None)).set_graph_mode(mode)
        assert step.comm.enabled and step.comm.world == world
        losses, first_g = [], None
        for it in range(6):
            img, cf, sdf, sal = (torch.from_numpy(a).cuda() for a in synth.make_batch(world * B, H, W, seed=40 + it))
            sl = slice(rank * B, (rank + 1) * B)
            losses.append(step.step(img[sl], cf[sl], sdf[sl], sal[sl]).cpu())
            if it == 0:
                first_g = step.flat_g.cpu()          # the first step's exchanged gradient: same weights in every variant
        torch.cuda.synchronize()
        ref = step.flat_p.clone()
        dist.broadcast(ref, 0)
        assert torch.equal(ref, step.flat_p), "ranks diverged"
        if mode != "off":
            assert step.graph_replays == 4, step.graph_replays                       # two eager warm-ups, then the chain
            caps = [c for c in step._graphs.values() if isinstance(c, graphs.StagedCaptured)]
            assert len(caps) == 1 and any(l == "scall" for l, _ in caps[0].segments) and any(l == "call" for l, _ in caps[0].segments)
        else:
            assert step.graph_replays == 0
        if rank == 0:
            # (compared HERE, against the first variant kept in host memory: three variants of dpt_large's 1.4-GB buffers written to
            # the test's temporary directory filled a GPU box's disk)
            cur = {"flat_p": step.flat_p.cpu(), "flat_g": step.flat_g.cpu(), "first_g": first_g, "losses": torch.stack(losses)}
            if mode == "off":
                keep = cur
            res = {"losses": cur["losses"], "same_losses": torch.equal(keep["losses"], cur["losses"]),
                   "same_flat_g": torch.equal(keep["flat_g"], cur["flat_g"]), "same_flat_p": torch.equal(keep["flat_p"], cur["flat_p"]),
                   "same_first_g": torch.equal(keep["first_g"], cur["first_g"])}
            a_, b_ = cur["first_g"].double(), keep["first_g"].double() / world
            res["first_g_rel_to_f32_mean"] = float((a_ - b_).norm() / b_.norm())
            res["first_g_cos_to_f32_mean"] = float(torch.dot(a_, b_) / (a_.norm() * b_.norm()))
            torch.save(res, os.path.join(out_dir, f"{mode}_{wire}.pt"))
            del cur
        del step, net
        torch.cuda.empty_cache()
        dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("backbone,tag,B,H,W,dtype_name", [("dpt_tiny", "tiny", 2, 64, 64, "float32"), ("dpt_large", "large", 10, 128, 128, "bfloat16")])
def test_data_parallel_step_replays_the_chain_of_graphs_bit_identically(tmp_path, backbone, tag, B, H, W, dtype_name):
    """Round 6: a data-parallel step of a small problem keeps the chain of per-stage graphs (trainer.TrainStep, graphs.StagedCaptured:
    the gradient all-reduces are host calls BETWEEN graph launches, at the bucket boundaries the chain is cut at, issued on the
    weight-gradient lane; finish() on the main lane before the optimizer).  Two ranks (gloo transport, CUDA tensors, one GPU) at the
    reference recipe's shape -- dpt_large, 2 x 10 crops of 128 x 128 (README.md:148-155) -- and on dpt_tiny in fp32: six steps replayed
    ('auto') against six eager steps ('off'): losses, exchanged gradients and weights bit-identical.  And the bf16 gradient wire
    against the f32 wire: same schedule, exchanged gradient within bf16 rounding."""
    world = 2
    mp.spawn(_worker_chain, args=(world, _free_port(), str(tmp_path), backbone, tag, B, H, W, dtype_name), nprocs=world, join=True)
    off, auto, wired = (torch.load(os.path.join(tmp_path, f"{m}_{w}.pt")) for m, w in (("off", "f32"), ("auto", "f32"), ("auto", "bf16")))
    assert auto["same_losses"] and auto["same_flat_g"] and auto["same_flat_p"] and auto["same_first_g"]
    # bf16 wire: the first step runs on the same weights in every variant -- same loss, and its exchanged gradient (already the mean over
    # ranks) equals the f32 exchange's sum / world to bf16 rounding: relative L2 <= 6e-3 (the bar of tests/test_parallel_cpu.py).  Later
    # steps are different trajectories (five different updates): finite, not compared
    assert torch.equal(off["losses"][0], wired["losses"][0])
    rel, cos = wired["first_g_rel_to_f32_mean"], wired["first_g_cos_to_f32_mean"]
    print(f"{backbone}: first-step gradient over the bf16 wire vs the f32 wire: relative L2 {rel:.2e}, cosine {cos:.7f}")
    assert rel <= 6e-3 and cos > 0.9999
    assert bool(torch.isfinite(wired["losses"]).all())


def _worker_rccl_one_rank(rank, world, port, out_dir, backbone, B, H, W, dtype_name):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    from argparse import Namespace
    from unmore_amd import graphs, synth
    from unmore_amd.objectness_net import ObjectnessNet
    from unmore_amd.trainer import TrainStep
    torch.manual_seed(0)
    init = {k: v.clone() for k, v in ObjectnessNet("cpu", H, backbone, Namespace(use_bg_sdf=True, sdf_activation="tanh")).state_dict().items()}
    warm = torch.ones(1 << 20, device="cuda:0")
    dist.all_reduce(warm)                        # the communicator is created by its first collective: outside every timed / compared region
    torch.cuda.synchronize()
    assert float(warm.sum()) == float(1 << 20)
    from unmore_amd import trainer as trainer_mod
    lag0 = trainer_mod._DP_ADAM_LAG
    keep = None
    for name, force, mode, wire in (("plain", "0", "off", None), ("rccl_eager", "1", "off", None), ("rccl_chain", "1", "auto", None),
                                    ("rccl_chain_traced", "1", "auto", None), ("rccl_chain_bf16", "1", "auto", torch.bfloat16),
                                    ("rccl_chain_lag1", "1", "auto", None), ("rccl_chain_lag100", "1", "auto", None)):
        os.environ["UMR_DP_FORCE"] = force
        # stages between a bucket's all-reduce and its optimizer launch (UMR_DP_ADAM_LAG): 1 = the very next stage, 100 = every stage after finish()
        trainer_mod._DP_ADAM_LAG = {"rccl_chain_lag1": 1, "rccl_chain_lag100": 100}.get(name, lag0)
        net = ObjectnessNet("cuda:0", H, backbone, Namespace(use_bg_sdf=True, sdf_activation="tanh"))
        net.load_state_dict(init, strict=True)
        net = net.to("cuda:0")
        net.set_compute_dtype(getattr(torch, dtype_name))
        step = TrainStep(net, lr=1e-4, grad_wire_dtype=wire).set_graph_mode(mode)
        assert step.comm.enabled == (force == "1") and step.comm.world == 1
        losses, first_g, trace = [], None, None
        for it in range(6):
            img, cf, sdf, sal = (torch.from_numpy(a).cuda() for a in synth.make_batch(B, H, W, seed=40 + it))
            if name == "rccl_chain_traced" and it == 4:
                step.comm.trace = True           # what bench.py does for one untimed step at world > 1: per-bucket issue / completion events
            losses.append(step.step(img, cf, sdf, sal).cpu())
            if step.comm.trace:
                trace = step.comm.trace_report()
                step.comm.trace = False
            if it == 0:
                first_g = step.flat_g.cpu()
        torch.cuda.synchronize()
        if mode != "off":
            assert step.graph_replays == 4, step.graph_replays
            caps = [c for c in step._graphs.values() if isinstance(c, graphs.StagedCaptured)]
            assert len(caps) == 1 and any(l == "scall" for l, _ in caps[0].segments) and any(l == "call" for l, _ in caps[0].segments)
            n_scall = sum(l == "scall" for l, _ in caps[0].segments)
            nb = step.comm.num_buckets
            if name == "rccl_chain_lag100":
                assert n_scall == nb, (n_scall, nb)                      # one ready() per bucket, every update after finish()
            elif name in ("rccl_chain", "rccl_chain_lag1"):
                assert n_scall > nb, (n_scall, nb)                       # + one wait() per stage that is updated on the weight-gradient lane
        if trace is not None:
            assert len(trace["buckets"]) == step.comm.num_buckets and all(r["done_ms"] >= r["issue_ms"] for r in trace["buckets"]), trace
        cur = {"flat_p": step.flat_p.cpu(), "flat_g": step.flat_g.cpu(), "first_g": first_g, "losses": torch.stack(losses)}
        if name == "plain":
            keep = cur          # compared in this process (seven variants of 1.4-GB buffers do not belong on the box's disk)
        a_, b_ = cur["first_g"].double(), keep["first_g"].double()
        torch.save({"losses": cur["losses"], "trace": trace, "same_losses": torch.equal(keep["losses"], cur["losses"]),
                    "same_flat_g": torch.equal(keep["flat_g"], cur["flat_g"]), "same_flat_p": torch.equal(keep["flat_p"], cur["flat_p"]),
                    "first_g_rel_to_plain": float((a_ - b_).norm() / b_.norm())}, os.path.join(out_dir, f"{name}.pt"))
        del cur
        del step, net
        torch.cuda.empty_cache()
    dist.destroy_process_group()


@pytest.mark.parametrize("backbone,B,H,W,dtype_name", [("dpt_tiny", 2, 64, 64, "float32"), ("dpt_large", 20, 128, 128, "bfloat16")])
def test_gradient_exchange_through_rccl_in_a_group_of_one_rank(tmp_path, backbone, B, H, W, dtype_name):
    """The exchange through the REAL backend.  RCCL refuses two ranks on one device, so the two-rank tests above use gloo, whose
    transport synchronises on the host: it cannot show a missing stream dependency around a collective.  A process group of one rank on
    the "nccl" backend (UMR_DP_FORCE=1: parallel.BucketedAllReduce exchanges even at world 1) takes every bucket through
    ProcessGroupNCCL: the collective's own stream ordered behind the issuing lane, work handles waited for on the main lane before the
    optimizer, the watchdog thread polling events while the chain of graphs is recorded, the per-bucket trace events of bench.py, the bf16
    wire's dtype.  A sum over one rank is the identity: six steps, eager and replayed from the chain of per-stage graphs, must equal the
    step without any exchange bit for bit; the bf16 wire differs by one rounding of the gradient."""
    mp.spawn(_worker_rccl_one_rank, args=(1, _free_port(), str(tmp_path), backbone, B, H, W, dtype_name), nprocs=1, join=True)
    r = {n: torch.load(os.path.join(tmp_path, f"{n}.pt")) for n in ("plain", "rccl_eager", "rccl_chain", "rccl_chain_traced", "rccl_chain_bf16",
                                                                    "rccl_chain_lag1", "rccl_chain_lag100")}
    for n in ("rccl_eager", "rccl_chain", "rccl_chain_traced", "rccl_chain_lag1", "rccl_chain_lag100"):
        assert r[n]["same_losses"] and r[n]["same_flat_g"] and r[n]["same_flat_p"], n
    print(backbone, "per-bucket trace of one replayed step through RCCL (world 1):", r["rccl_chain_traced"]["trace"])
    rel = r["rccl_chain_bf16"]["first_g_rel_to_plain"]
    assert torch.equal(r["plain"]["losses"][0], r["rccl_chain_bf16"]["losses"][0])
    assert rel <= 3e-3, rel                      # one rounding to bf16 (uniform relative error <= 2^-9: rms 2^-9 / sqrt 3 = 1.1e-3)
    assert bool(torch.isfinite(r["rccl_chain_bf16"]["losses"]).all())
