"""crops/s of the sdf-only forward against the batch size (the boundary rounds' batch: object_reasoning.py:397 uses 50)"""
import os, sys, time, json
sys.path.insert(0, os.getcwd())
import torch
from argparse import Namespace
from unmore_amd.objectness_net import ObjectnessNet
dev = "cuda:0"
backbone = sys.argv[1] if len(sys.argv) > 1 else "dpt_large"
torch.manual_seed(0)
net = ObjectnessNet(dev, 128, backbone, Namespace(use_bg_sdf=True, sdf_activation="tanh")).to(dev).eval()
for p in net.parameters():
    p.requires_grad = False
for dt in (torch.float32, torch.bfloat16):
    net.set_compute_dtype(dt)
    for B in (25, 50, 100, 200, 400):
        x = torch.rand(B, 3, 128, 128, device=dev)
        with torch.no_grad():
            for _ in range(4):
                net.get_prediction(x, heads=("sdf_maps",))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 10
            for _ in range(n):
                net.get_prediction(x, heads=("sdf_maps",))
            torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        print(json.dumps({"backbone": backbone, "dtype": str(dt).split(".")[-1], "batch": B, "ms": round(ms, 2), "crops_per_s": round(B / ms * 1e3)}), flush=True)
