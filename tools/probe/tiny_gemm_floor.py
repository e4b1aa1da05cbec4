"""Fixed cost of one launch of the 128x128 NT kernel: device time of a 12-tile GEMM as K shrinks to one K-tile.
python tools/probe/tiny_gemm_floor.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unmore_amd import ops
dev = torch.device("cuda:0")
ops.set_debug_option("UMR_NT_SPLITK", "0")


def bench(fn, n=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for K in (64, 128, 384, 768, 1536):
    A = torch.randn(394, K, device=dev).bfloat16()
    B = torch.randn(384, K, device=dev).bfloat16()
    bias = torch.zeros(384, device=dev)
    out = torch.empty(394, 384, device=dev, dtype=torch.bfloat16)
    print(f"M=394 N=384 K={K:5d} ({K // 64:2d} K-tiles, 12 workgroups): {bench(lambda: ops.gemm_nt(A, B, bias, out=out)):6.1f} us per launch (back-to-back launches, events)", flush=True)
x = torch.randn(394, 384, device=dev).bfloat16()
print(f"cast of the same 394 x 384 tensor (a trivial kernel): {bench(lambda: ops.cast(x, torch.float32)):6.1f} us per launch")
