import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from argparse import Namespace
from unmore_amd import synth
from unmore_amd.objectness_net import ObjectnessNet
from unmore_amd.trainer import TrainStep
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = ObjectnessNet(dev, 384, "dpt_base", Namespace(use_bg_sdf=True, sdf_activation="tanh")).to(dev)
net.set_compute_dtype(torch.bfloat16); net.train()
step = TrainStep(net, lr=1e-4)
img, cf, sdf, sal = (torch.from_numpy(x).to(dev) for x in synth.make_batch(64, 384, 384, seed=0))
for _ in range(2): step.step(img, cf, sdf, sal)
torch.cuda.synchronize()
ts = []
for _ in range(4):
    t0 = time.perf_counter(); step.step(img, cf, sdf, sal); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    ts.append((t1 - t0, t2 - t0))
print("host enqueue ms / full step ms:", [(round(a * 1e3, 1), round(b * 1e3, 1)) for a, b in ts])
