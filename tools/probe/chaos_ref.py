"""Is the step-13 split between the bf16 and fp32 loss trajectories of the reference recipe (tools/train_sanity.py, TRAIN_SANITY_CFG=ref)
the training's own sensitivity?  Two fp32-grade runs (exact f32 MFMA vs three-way bf16 split products, ~1e-7 apart per product) from
the same weights on the same batches: python tools/probe/chaos_ref.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from argparse import Namespace
import torch
from unmore_amd import synth, ops
from unmore_amd.objectness_net import ObjectnessNet
from unmore_amd.trainer import TrainStep

dev = torch.device("cuda:0")
torch.manual_seed(0)
net0 = ObjectnessNet(dev, 128, "dpt_large", Namespace(use_bg_sdf=True, sdf_activation="tanh")).to(dev)
init = {k: v.detach().clone() for k, v in net0.state_dict().items()}
del net0
pool = []
for b in range(4):
    _, cf, sdf, sal = synth.make_batch(20, 128, 128, seed=100 + b)
    img = synth.blob_images(20, 128, 128, seed=100 + b)
    pool.append(tuple(torch.from_numpy(a).to(dev) for a in (img, cf, sdf, sal)))
res = {}
for mode in ("exact", "x3"):
    ops.set_f32_mode(mode)
    net = ObjectnessNet(dev, 128, "dpt_large", Namespace(use_bg_sdf=True, sdf_activation="tanh")).to(dev)
    net.load_state_dict(init, strict=True)
    net.set_compute_dtype(torch.float32)
    net.train()
    step = TrainStep(net, lr=1e-4, lr_milestones=(10000, 20000), lr_gamma=0.1)
    res[mode] = [round(float(step.step(*pool[it % 4])[0]), 4) for it in range(24)]
    del step, net
    print(mode, res[mode], flush=True)
d = [abs(a - b) for a, b in zip(res["exact"], res["x3"])]
print("|exact - x3| per step:", [round(x, 4) for x in d])
