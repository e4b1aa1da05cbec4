"""Tile order of the 128x128 NT kernel inside an XCD's run (UMR_NT_ORDER=n|m: ops.set_debug_option): the reference recipe's GEMM shapes (1300 tokens, ViT-L),
weights rotated through a pool larger than the infinity cache (as in the step, where every layer's weights arrive cold), alternating orders.
python tools/probe/nt_order_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unmore_amd import ops

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)


def bench(fn, n=120):
    for i in range(8):
        fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n):
        fn(i)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for (M, N, K) in [(1300, 4096, 1024), (1300, 3072, 1024), (1300, 1024, 1024), (1300, 1024, 4096), (1300, 1024, 3072), (3250, 2304, 768), (3250, 3072, 768), (394, 1536, 384)]:
    pool = max(2, int(400e6 // (N * K * 2)))
    A = torch.randn((M, K), generator=g).to(dev).bfloat16()
    Bs = [(torch.randn((N, K), generator=g) * K ** -0.5).to(dev).bfloat16() for _ in range(min(pool, 64))]
    bias = torch.zeros(N, device=dev)
    out = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
    res = {"n": [], "m": []}
    for rnd in range(3):
        for order in ("n", "m"):
            ops.set_debug_option("UMR_NT_ORDER", order)
            res[order].append(bench(lambda i: ops.gemm_nt(A, Bs[i % len(Bs)], bias, out=out)))
    ops.set_debug_option("UMR_NT_ORDER", None)
    auto = bench(lambda i: ops.gemm_nt(A, Bs[i % len(Bs)], bias, out=out))
    print(f"M={M} N={N} K={K} ({len(Bs)} weight sets)  us per launch  n-fastest {[round(v, 1) for v in res['n']]}  m-fastest {[round(v, 1) for v in res['m']]}  default {auto:.1f}", flush=True)
