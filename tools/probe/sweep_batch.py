"""reasoning.sweep_proposals (certified fp32, three streams) against its batch size: 1225 proposals of a 640x480 image, ViT-B wiring with
the peak-producing weights of the test suite"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from argparse import Namespace
import bench
from unmore_amd import reasoning, synth
from unmore_amd.objectness_net import ObjectnessNet
dev = "cuda:0"
net = ObjectnessNet(dev, 128, "dpt_base", Namespace(use_bg_sdf=True, sdf_activation="tanh"))
spec = {k: tuple(v.shape) for k, v in net.state_dict().items()}
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.peak_edited_state_dict(spec, "base").items()}, strict=True)
net = net.to(dev).eval()
for p in net.parameters():
    p.requires_grad = False
image = torch.from_numpy(synth.blob_images(1, 480, 640, seed=5)[0]).to(dev)
props = torch.from_numpy(bench.anchors(480, 640))
for dt in (torch.float32, torch.bfloat16):
    net.set_compute_dtype(dt)
    ref = None
    for nb in (50, 100, 200):
        info = {}
        for _ in range(3):
            out = reasoning.sweep_proposals(net, image, props, num_img_per_batch=nb, precision="certified", info=info)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            out = reasoning.sweep_proposals(net, image, props, num_img_per_batch=nb, precision="certified", info=info)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        if ref is None:
            ref = out
        same = int((out[1] != ref[1]).sum())
        print(json.dumps({"dtype": str(dt).split(".")[-1], "batch": nb, "ms_per_image": round(ms, 1), "rerun": info.get("rerun"), "peak_indices_differing_from_batch_50": same}), flush=True)
