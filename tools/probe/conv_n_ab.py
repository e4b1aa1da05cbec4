"""Does the second N-tile's re-read of the A operand cost the head conv anything?  3x3 conv 512 -> N at the cfg2 shape for N = 256
(one N-tile: every A tile is staged by exactly one workgroup), 512 (two: the benchmark's layer) and 1024 (four), interleaved
rounds in one process: if TFLOP/s falls with N, A-sharing across workgroups matters.  python tools/probe/conv_n_ab.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from unmore_amd import ops, _lib as L
from kbench import timeit

dev = torch.device("cuda:0")
B, H, W = 64, 384, 384
M = B * H * W
g = torch.Generator().manual_seed(0)
x = torch.randn((B, H, W, 512), generator=g).to(dev).to(torch.bfloat16)
res = {}
ws = {N: (torch.randn((N, 4608), generator=g) * 0.02).to(dev).to(torch.bfloat16) for N in (256, 512, 1024)}
outs = {N: torch.empty((M, N), dtype=torch.bfloat16, device=dev) for N in (256, 512, 1024)}
bias = {N: torch.zeros(N, device=dev) for N in (256, 512, 1024)}
for rnd in range(3):
    for N in (256, 512, 1024):
        t = timeit(lambda: ops.gemm_nt(x, ws[N], bias[N], conv=1, act=L.ACT_RELU, out=outs[N]), n=5, warm=2)
        res.setdefault(N, []).append(t)
for N in (256, 512, 1024):
    t = min(res[N])
    print(f"conv3x3 512->{N:4d}: {t:8.3f} ms  {2.0 * M * N * 4608 / t / 1e9:7.1f} TFLOP/s", flush=True)
