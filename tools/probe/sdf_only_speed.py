"""What the boundary-reasoning rounds (object_reasoning.py:379-487) save by not evaluating the centre head: get_prediction on a
[50,3,128,128] batch, full against heads=("sdf_maps",), ViT-B/16 wiring and dpt_large, bf16 and fp32 (three-plane products), replayed
from the per-shape graphs.   python tools/probe/sdf_only_speed.py"""
import os, sys, time, json
sys.path.insert(0, os.getcwd())
import torch
from argparse import Namespace
from unmore_amd import ops
from unmore_amd.objectness_net import ObjectnessNet

dev = "cuda:0"
x = torch.rand(50, 3, 128, 128, device=dev)
for backbone in ("dpt_base", "dpt_large"):
    torch.manual_seed(0)
    net = ObjectnessNet(dev, 128, backbone, Namespace(use_bg_sdf=True, sdf_activation="tanh")).to(dev).eval()
    for p in net.parameters():
        p.requires_grad = False
    for dt, mode in ((torch.bfloat16, None), (torch.float32, "x3"), (torch.float32, "x3_fast")):
        net.set_compute_dtype(dt)
        if mode:
            ops.set_f32_mode(mode)
        row = {"backbone": backbone, "dtype": str(dt).split(".")[-1] + (f" ({mode})" if mode else "")}
        for name, heads in (("full", None), ("sdf_only", ("sdf_maps",)), ("center_only", ("center_fields",))):
            with torch.no_grad():
                for _ in range(5):
                    net.get_prediction(x, heads=heads)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(20):
                    net.get_prediction(x, heads=heads)
                torch.cuda.synchronize()
            row[name + "_ms"] = round((time.perf_counter() - t0) / 20 * 1e3, 3)
        row["full_over_sdf_only"] = round(row["full_ms"] / row["sdf_only_ms"], 2)
        print(json.dumps(row), flush=True)
    ops.set_f32_mode("x3")
