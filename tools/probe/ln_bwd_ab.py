"""LayerNorm backward at the cfg2 token count (36928 x 768 bf16): python tools/probe/ln_bwd_ab.py  (UMR_LIB / UMR_LN_BWD_WG_PER_CU vary)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unmore_amd import ops
from tools.kbench import timeit
dev = torch.device("cuda:0")
for (M, D) in ((64 * 577, 768), (16 * 1370, 1024), (1300, 1024)):
    x = torch.randn(M, D, device=dev).bfloat16()
    dy = torch.randn(M, D, device=dev).bfloat16()
    res = torch.randn(M, D, device=dev).bfloat16()
    g, b = torch.ones(D, device=dev), torch.zeros(D, device=dev)
    _, mu, rs = ops.layernorm_fwd(x, g, b)
    dg, db = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
    t = timeit(lambda: ops.layernorm_bwd(dy, x, g, mu, rs, dg, db, dres=res), n=50)
    print(f"M={M} D={D}: {t * 1e3:7.1f} us  {4 * M * D * 2 / t / 1e9:5.2f} TB/s", flush=True)
