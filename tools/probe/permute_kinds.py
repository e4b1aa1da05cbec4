"""Bandwidth of the batched weight refresh by entry class: plain casts ('lin') and 64x64 transposes ('lin_t') of the ViT-L Linear weights."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unmore_amd import engine, ops

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
ws = []
for _ in range(24):
    for n, k in ((3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096)):
        ws.append(torch.randn((n, k), generator=g).to(dev))
def recs(fn):
    out = []
    for w in ws:
        rec = []
        ops._pack_recorder = rec
        try:
            fn(w)
        finally:
            ops._pack_recorder = None
        out.append(rec[0])
    return out
nel = sum(w.numel() for w in ws)
for name, fn in (("casts", lambda w: engine._pack_linear(w, torch.bfloat16)), ("transposes", lambda w: engine._pack_linear_t(w, torch.bfloat16))):
    launch = ops.permute4_batched(recs(fn))
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        launch()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    print(f"{name:11s} {nel / 1e6:6.1f} M elements  {ms:6.3f} ms  {nel * 6 / ms / 1e9:6.2f} TB/s (4 B read + 2 B written per element)")
