"""Time of the per-step refresh of every kernel-layout weight copy (PackCache.refresh = one umr_permute4_batched launch per stage) of dpt_large
in bf16, as the reference recipe's step runs it: python tools/probe/refresh_ab.py   (A/B: UMR_LIB=<other library>)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from argparse import Namespace
import torch
from unmore_amd.objectness_net import ObjectnessNet
from unmore_amd import trainer

dev = torch.device("cuda:0")
torch.manual_seed(0)
net = ObjectnessNet(dev, 128, "dpt_large", Namespace(use_bg_sdf=True, sdf_activation="tanh")).to(dev)
net.set_compute_dtype(torch.bfloat16)
step = trainer.TrainStep(net, lr=1e-4).set_graph_mode("off")
g = torch.Generator().manual_seed(1)
B = 20
batch = (torch.randn(B, 3, 128, 128, generator=g).to(dev), torch.randn(B, 2, 128, 128, generator=g).to(dev) * 0.1,
         torch.rand(B, 1, 128, 128, generator=g).to(dev), (torch.rand(B, 1, 128, 128, generator=g) > 0.5).float().to(dev))
for _ in range(3):
    step.step(*batch)
torch.cuda.synchronize()
cache = net._engine().cache
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10):
    cache.refresh()
b.record()
torch.cuda.synchronize()
print(f"full refresh of {len(cache._c)} weight copies: {a.elapsed_time(b) / 10:7.3f} ms")
