"""Where the fixed per-tile cost of the persistent 256x256 NT kernel goes: same GEMM with and without the C store
(no_store keeps the epilogue math, LDS staging and barriers), at a few K.  Run on the MI355X: python tools/epi_cost.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from unmore_amd import ops, _lib as L
from kbench import timeit

dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
M = 64 * 384 * 384 // 4
a = (torch.randn(M // 64, 2048, generator=g)).to(dev).bfloat16().repeat(64, 1)
for K, N in ((256, 512), (768, 512), (768, 2304), (2048, 512)):
    A = a[:, :K].contiguous()
    w = (torch.randn(N, K, generator=g) * 0.03).to(dev).bfloat16()
    bias = torch.zeros(N, device=dev)
    rw = torch.ones(1, N, device=dev)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    tiles = ((M + 255) // 256) * ((N + 255) // 256) / 256.0
    t1 = timeit(lambda: ops.gemm_nt(A, w, bias, act=L.ACT_RELU, out=out))
    t2 = timeit(lambda: ops.gemm_nt(A, w, bias, act=L.ACT_RELU, red_w=rw, no_store=True))
    t3 = timeit(lambda: ops.gemm_nt(A, w, bias, act=L.ACT_RELU, red_w=rw, out=out))
    t4 = timeit(lambda: ops.gemm_nt(A, w, None, act=L.ACT_RELU, out=out))
    t5 = timeit(lambda: ops.gemm_nt(A, w, None, out=out))
    print(f"K={K:5d} N={N:5d}: store {t1 * 1e3 / tiles:6.2f} us/tile   no_store(+red) {t2 * 1e3 / tiles:6.2f}   store+red {t3 * 1e3 / tiles:6.2f}   "
          f"no bias {t4 * 1e3 / tiles:6.2f}   no bias, no act {t5 * 1e3 / tiles:6.2f}   ({K // 64} k-tiles; C tile = 128 KiB)", flush=True)
