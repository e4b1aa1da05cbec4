"""Throughput of the on-device ground-truth synthesis at the training batch shape: python tools/labels_bench.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from unmore_amd.labels import synthesize_labels
from unmore_amd.synth import ellipse_masks

B, H, W = 64, 384, 384
m = torch.from_numpy(np.asarray(ellipse_masks(B, H, W, seed=0)).astype(np.uint8)).cuda()
for _ in range(2):
    synthesize_labels(m)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 10
for _ in range(n):
    out = synthesize_labels(m)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"label synthesis B={B} {H}x{W}: {1e3 * dt:.2f} ms per batch = {B / dt:.0f} images/s (sdf range {float(out['sdf'].min()):.3f}..{float(out['sdf'].max()):.3f})")

# the random-crop branch of the training item (datasets.py:144-190): decoded 500x375 images -> 400x400 -> crop -> 384x384
from unmore_amd.labels import synthesize_training_items
g = torch.Generator().manual_seed(0)
imgs = [torch.rand(3, 375, 500, device="cuda") for _ in range(B)]
mks = [torch.from_numpy(np.asarray(ellipse_masks(1, 375, 500, seed=i))[0].astype(np.float32)).cuda() for i in range(B)]
for _ in range(2):
    synthesize_training_items(imgs, mks, 384, generator=g)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    synthesize_training_items(imgs, mks, 384, generator=g)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"training items (resize 400 + DT + random crop + resize 384 + labels) B={B}: {1e3 * dt:.2f} ms per batch = {B / dt:.0f} images/s")
