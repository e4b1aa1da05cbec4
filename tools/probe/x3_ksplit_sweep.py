"""K-split cap sweep of the plane GEMMs at the reference recipe's shapes: one process per cap (UMR_X3_KSPLIT is read once).
    for k in 1 2 3 4 6 8 12 16 32; do UMR_X3_KSPLIT=$k python tools/probe/x3_ksplit_sweep.py; done"""
import os
import sys
import torch
sys.path.insert(0, ".")
from unmore_amd import ops, _lib as L

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
out = [f"cap {os.environ.get('UMR_X3_KSPLIT', 'model'):>5s}"]
def t(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for name, M, N, K in (("qkv", 1300, 3072, 1024), ("proj", 1300, 1024, 1024), ("fc1", 1300, 4096, 1024), ("fc2", 1300, 1024, 4096), ("b-qkv", 3250, 2304, 768), ("b-fc2", 3250, 768, 3072)):
    Ap, Bp, bias = ops.split3(rnd(M, K)), ops.split3(rnd(N, K) * K ** -0.5), rnd(N)
    out.append(f"{name} {t(lambda: ops.gemm_nt_x3(Ap, Bp, bias)):6.1f}")
for name, nb, H, W, Cin, N in (("c8", 20, 8, 8, 256, 256), ("c16", 20, 16, 16, 256, 256), ("c32", 20, 32, 32, 256, 256), ("c64", 20, 64, 64, 256, 256)):
    xp, wp, bias = ops.split3(rnd(nb, H, W, Cin)), ops.split3(rnd(N, 9 * Cin) * 0.02), rnd(N)
    out.append(f"{name} {t(lambda: ops.gemm_nt_x3(xp, wp, bias, conv=1, act=L.ACT_RELU), 10):6.1f}")
print("  ".join(out))
