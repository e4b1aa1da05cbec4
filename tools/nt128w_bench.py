"""A/B of the two-workgroups-per-CU NT kernel (gemm_nt128w.hip) against the persistent 256x256 kernel on the short-K GEMMs of the
step: the heads' 1x1 layers at cfg2 (M = 64 x 384 x 384) and the ViT-B GEMMs at 36,928 tokens.  Interleaved rounds in one process
(UMR_NT128W is read per launch).  python tools/nt128w_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from unmore_amd import ops, _lib as L
from kbench import timeit

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
rnd = lambda *s, scale=1.0, dt=torch.bfloat16: (torch.randn(s, generator=g) * scale).to(dev).to(dt)
Mh, Mt = 64 * 384 * 384, 64 * 577
cases = []


def add(name, M, N, K, **kw):
    x, w = rnd(M, K), rnd(N, K, scale=K ** -0.5)
    bias = torch.zeros(N, device=dev)
    out = None if kw.get("no_store") else torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    if kw.get("aux_kind"):
        aux = rnd(M, N)
        kk = dict(aux=aux, mask_relu=(kw["aux_kind"] == "mask"))
    else:
        kk = {}
    if kw.get("red"):
        kk.update(red_w=rnd(2, N, dt=torch.float32), no_store=bool(kw.get("no_store")))
    fn = lambda: ops.gemm_nt(x, w, None if kw.get("aux_kind") == "mask" else bias, act=kw.get("act", L.ACT_NONE), out=out, **kk)
    cases.append((name, 2.0 * M * N * K, fn))


add("head 256->512 relu", Mh, 512, 256, act=L.ACT_RELU)
add("head 512->1024 relu + fused out", Mh, 1024, 512, act=L.ACT_RELU, red=True)
add("head 512->1024 + fused out, no store", Mh, 1024, 512, red=True, no_store=True)
add("head 1024->512 masked dgrad", Mh, 512, 1024, aux_kind="mask")
add("head 1024->512 dgrad (no mask)", Mh, 512, 1024)
add("head 1024->256 dfeat", Mh, 256, 1024)
add("vit qkv", Mt, 2304, 768)
add("vit proj (+res)", Mt, 768, 768, aux_kind="add")
add("vit fc1 dgrad", Mt, 768, 3072)
for name, fl, fn in cases:
    res = {}
    for rnd_ in range(3):
        for mode in ("0", "2", "3"):
            os.environ["UMR_NT128W"] = mode
            res.setdefault(mode, []).append(timeit(fn, n=7, warm=2))
    a, b, c = min(res["0"]), min(res["2"]), min(res["3"])
    print(f"{name:40s} 256p {a:8.3f} ms {fl / a / 1e9:7.1f} TF/s | 128x256 {b:8.3f} ms {100 * (a / b - 1):+5.1f} % | 128x512 {c:8.3f} ms {fl / c / 1e9:7.1f} TF/s {100 * (a / c - 1):+5.1f} %", flush=True)
