"""The plane (fp32-grade) GEMMs of ViT-L at the cfg5 sweep's token count (50 crops x 65 tokens): GEMM + K-split finish, per shape.
Run under rocprofv3 --kernel-trace --stats to see the two kernels apart; prints the pair's time and the K-split the plan chose."""
import sys
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from unmore_amd import ops, _lib as L

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
M = 3250
res = rnd(M, 1024)
for name, N, K, kw in (("qkv", 3072, 1024, {}), ("proj + residual", 1024, 1024, dict(aux=res)), ("fc1 + GELU (pre saved)", 4096, 1024, dict(act=L.ACT_GELU)),
                       ("fc2 + residual", 1024, 4096, dict(aux=res))):
    Ap, Bp, bias = ops.split3(rnd(M, K)), ops.split3(rnd(N, K) * K ** -0.5), rnd(N)
    for planes in (False, True):
        us = t(lambda: ops.gemm_nt_x3(Ap, Bp, bias, out_planes=planes, **kw))
        print(f"{name:24s} N={N:5d} K={K:5d} out {'planes' if planes else 'f32   '}: {us:7.1f} us")
