"""The two kernels of the algebraic boundary-distance-head backward at the cfg2 shape: python tools/lh_bench.py [B]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from unmore_amd import ops, _lib as L
from tools.kbench import timeit

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
H = W = 384
dev = torch.device("cuda:0")
x = torch.randn(B, H, W, 256, device=dev).bfloat16()
dout = torch.randn(B, 1, H, W, device=dev)
yout = torch.tanh(torch.randn(B, 1, H, W, device=dev))
kw = torch.randn(9, 256, device=dev)
dx = torch.zeros(B, H, W, 256, device=dev, dtype=torch.bfloat16)
nbytes = x.numel() * 2
t = timeit(lambda: ops.linear_head_bwd_weight(x, dout, yout, L.ACT_TANH))
print(f"lh_bwd_weight: {t:.3f} ms  {nbytes / t / 1e9:.2f} TB/s")
t = timeit(lambda: ops.linear_head_bwd_data(dout, yout, kw, dx, L.ACT_TANH, True))
print(f"lh_bwd_data (accumulate): {t:.3f} ms  {2 * nbytes / t / 1e9:.2f} TB/s")
