"""cfg5-style inference: [50,3,128,128] crops through ObjectnessNet.get_prediction under no_grad."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from argparse import Namespace
from unmore_amd.objectness_net import ObjectnessNet

backbone = sys.argv[1] if len(sys.argv) > 1 else "dpt_base"
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = ObjectnessNet(dev, 128, backbone, Namespace(use_bg_sdf=True, sdf_activation="tanh")).to(dev).eval()
for p in net.parameters():
    p.requires_grad = False
x = torch.rand(NB, 3, 128, 128, device=dev)
for dt in (torch.bfloat16, torch.float32):
    net.set_compute_dtype(dt)
    with torch.no_grad():
        for _ in range(3):
            net.get_prediction(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            out = net.get_prediction(x)
        torch.cuda.synchronize()
        dtm = (time.perf_counter() - t0) / n
    print(f"{backbone} {dt}: {dtm*1e3:.2f} ms per {NB}-crop forward = {NB/dtm:.0f} crops/s")
