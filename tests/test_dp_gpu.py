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


def _make_net():
    from unmore_amd.hashrng import hash_init
    from unmore_amd.objectness_net import ObjectnessNet
    net = ObjectnessNet("cuda:0", 64, "dpt_tiny", ARGS)
    sd = {k: torch.from_numpy(hash_init(k, tuple(v.shape), "tiny")) for k, v in net.state_dict().items()}
    net.load_state_dict(sd, strict=True)
    return net.to("cuda:0")


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from unmore_amd import synth
    from unmore_amd.trainer import TrainStep
    net = _make_net()
    img, cf, sdf, sal = (torch.from_numpy(a).cuda() for a in synth.make_batch(4, 64, 64, seed=7))
    sl = slice(rank * 2, rank * 2 + 2)
    step = TrainStep(net, lr=1e-3)
    assert step.comm.enabled and step.comm.world == 2
    step.step(img[sl], cf[sl], sdf[sl], sal[sl])
    torch.cuda.synchronize()
    torch.save({k: v.detach().cpu() for k, v in net.state_dict().items()}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_step_equals_single_process_on_global_batch(tmp_path):
    from unmore_amd import synth
    from unmore_amd.trainer import TrainStep
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(os.path.join(tmp_path, "r0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "r1.pt"))
    for k in r0:
        assert torch.equal(r0[k], r1[k]), f"ranks diverged on {k}"
    net = _make_net()
    img, cf, sdf, sal = (torch.from_numpy(a).cuda() for a in synth.make_batch(4, 64, 64, seed=7))
    TrainStep(net, lr=1e-3).step(img, cf, sdf, sal)
    ref = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    bad = 0
    total = 0
    for k in ref:
        # Adam's first step is ~lr*sign(g): only weights whose gradient is ~0 may land on the other side
        bad += int(((ref[k] - r0[k]).abs() > 2e-4).sum())
        total += ref[k].numel()
    assert bad <= 2e-3 * total, (bad, total)
