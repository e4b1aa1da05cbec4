"""The 512-channel resizes of the heads' first layer at cfg2 (192^2 <-> 384^2, B = 64): forward with the fused ReLU, adjoint into a
column slice, for a few grid caps.   python tools/probe/bilinear512.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from unmore_amd import ops
from tools.kbench import timeit

dev = torch.device("cuda:0")
x = torch.randn(64, 192, 192, 512, device=dev).bfloat16()
dy = torch.randn(16, 384, 384, 512, device=dev).bfloat16().repeat(4, 1, 1, 1)
dlow = torch.empty((64 * 192 * 192, 576), dtype=torch.bfloat16, device=dev)
ref = None
for cap in (None, "8192", "4096", "1024", "512"):
    if cap is None:
        ops.set_debug_option("UMR_BILINEAR_GY", None)
    else:
        ops.set_debug_option("UMR_BILINEAR_GY", cap)
    tf = timeit(lambda: ops.bilinear_fwd(x, 384, 384, True, relu=True), n=7, warm=2)
    tb = timeit(lambda: ops.bilinear_bwd(dy, 192, 192, True, out=dlow[:, :512].unflatten(0, (64, 192, 192))), n=7, warm=2)
    r = dlow[:, :512].float().sum().item()
    ref = r if ref is None else ref
    print(f"gy cap {cap or 'dflt':>5s}: fwd {tf:6.3f} ms ({12.08 / tf:5.2f} TB/s)   adjoint {tb:6.3f} ms ({12.08 / tb:5.2f} TB/s)   checksum diff {abs(r - ref):.3g}", flush=True)
ops.set_debug_option("UMR_BILINEAR_GY", None)
