"""Captured train steps over a changing batch size (the reference's batch filter, train_objectness_net.py:190-207): every shape is captured on its
third step, at most graphs.MAX_CAPTURES captures are held -- reserved memory must level off.  python tools/probe/capture_memory.py"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from argparse import Namespace
from unmore_amd import graphs, synth
from unmore_amd.objectness_net import ObjectnessNet
from unmore_amd.trainer import TrainStep
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = ObjectnessNet(dev, 128, "dpt_base", Namespace(use_bg_sdf=True, sdf_activation="tanh")).to(dev)
net.set_compute_dtype(torch.bfloat16); net.train()
step = TrainStep(net, lr=1e-4)
for rnd in range(3):
    for B in range(6, 20):
        batch = tuple(torch.from_numpy(a).to(dev) for a in synth.make_batch(B, 128, 128, seed=B))
        for _ in range(4):
            loss = step.step(*batch)
        torch.cuda.synchronize()
    caps = sum(isinstance(v, graphs.CAPTURE_TYPES) for v in step._graphs.values())
    print(f"round {rnd}: captures held {caps}, replays {step.graph_replays}, allocated {torch.cuda.memory_allocated() / 2**30:.2f} GiB, "
          f"reserved {torch.cuda.memory_reserved() / 2**30:.2f} GiB, loss {loss[0].item():.4f}", flush=True)
