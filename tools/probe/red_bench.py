"""The 512 -> 1024 head GEMM with the fused 1024 -> 2 output layer at the cfg2 size (M = 64 x 384^2): stored and no_store forms."""
import sys
import torch
sys.path.insert(0, ".")
from unmore_amd import ops, _lib as L

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
M, K, N = 64 * 384 * 384, 512, 1024
A = torch.randn(M // 64, K, generator=g).to(dev).bfloat16().repeat(64, 1)
w = (torch.randn(N, K, generator=g) * 0.04).to(dev).bfloat16()
bias = torch.zeros(N, device=dev)
rw = torch.randn(2, N, generator=g).to(dev)
out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
for name, kw in (("fused, h3 stored", dict(out=out, red_w=rw)), ("fused, no_store", dict(red_w=rw, no_store=True)), ("plain (no reduction)", dict(out=out))):
    fn = lambda: ops.gemm_nt(A, w, bias, act=L.ACT_RELU, **kw)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        fn()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    print(f"{name:24s} {ms:7.3f} ms  {2.0 * M * N * K / ms / 1e9:7.1f} TFLOP/s")
