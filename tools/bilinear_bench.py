"""The final x2 resize of the feature map at cfg2 (192^2 -> 384^2, 256 channels, B = 64, align_corners) forward and adjoint, for
several grid caps (UMR_BILINEAR_GY, read per launch).  python tools/bilinear_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from unmore_amd import ops
from kbench import timeit

dev = torch.device("cuda:0")
x = torch.randn(64, 192, 192, 256, device=dev).bfloat16()
dy = torch.randn(64, 384, 384, 256, device=dev).bfloat16()
for cap in (None, "0", "8192", "4096", "2048", "1024", "512", None):      # None = the library's defaults
    if cap is None:
        ops.set_debug_option("UMR_BILINEAR_GY", None)
    else:
        ops.set_debug_option("UMR_BILINEAR_GY", cap)
    cap = cap or "dflt"
    tf = timeit(lambda: ops.bilinear_fwd(x, 384, 384, True), n=9, warm=2)
    tb = timeit(lambda: ops.bilinear_bwd(dy, 192, 192, True), n=9, warm=2)
    print(f"gy cap {cap:>5s}: fwd {tf:6.3f} ms ({6.04 / tf:5.2f} TB/s)   bwd {tb:6.3f} ms ({6.04 / tb:5.2f} TB/s)", flush=True)
