"""Adam over the flat buffers of dpt_large (343 M parameters): time and HBM rate (16 B read + 12 B written per parameter)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unmore_amd import ops
dev = torch.device("cuda:0")
n = 343_000_000 // 64 * 64
p, g, m, v = (torch.randn(n, device=dev) * 0.01 for _ in range(4))
v.abs_()
hyper = torch.zeros(8, dtype=torch.float32, device=dev)
ops.adam_set_hyper(hyper, 3, 1e-4, 0.9, 0.999, 1e-8, 1.0)
for parts in (1, 28):
    bounds = [n * i // parts // 64 * 64 for i in range(parts)] + [n]
    fn = lambda: [ops.adam_step_hyper(p[a:b], g[a:b], m[a:b], v[a:b], hyper) for a, b in zip(bounds[:-1], bounds[1:])]
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): fn()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    print(f"{parts:3d} launches: {ms:6.3f} ms  {n * 28 / ms / 1e9:5.2f} TB/s")
