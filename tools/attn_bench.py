"""Attention kernels at the cfg2 ViT-B shape (B=64, N=577, 12 heads, head dim 64): python tools/attn_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from unmore_amd import ops
from kbench import timeit

B, N, H = (int(v) for v in os.environ.get("ATTN_SHAPE", "64,577,12").split(","))     # cfg4: 16,1370,16   ref recipe: 20,65,16
D = H * 64
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
qkv = (torch.randn(B * N, 3 * D, generator=g) * 0.5).to(dev).bfloat16()
dout = torch.randn(B * N, D, generator=g).to(dev).bfloat16()
out, lse = ops.attention_fwd(qkv, B, N, H)
t = timeit(lambda: ops.attention_fwd(qkv, B, N, H), n=20)
fl = 4.0 * B * H * N * N * 64
print(f"attention fwd : {t:7.3f} ms  {fl / t / 1e9:7.1f} TFLOP/s")
t = timeit(lambda: ops.attention_bwd(qkv, out, dout, lse, B, N, H), n=20)
print(f"attention bwd : {t:7.3f} ms  {2.5 * fl / t / 1e9:7.1f} TFLOP/s (10 N^2 D flops)")

if os.environ.get("ATTN_ONLY"):
    sys.exit(0)
# LayerNorm at the same token count
x = torch.randn(B * N, D, generator=g).to(dev).bfloat16()
gam, bet = torch.ones(D, device=dev), torch.zeros(D, device=dev)
y, mu, rs = ops.layernorm_fwd(x, gam, bet)
t = timeit(lambda: ops.layernorm_fwd(x, gam, bet), n=20)
print(f"layernorm fwd : {t:7.3f} ms  {2 * x.numel() * 2 / t / 1e9:6.2f} TB/s")
dg, db = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
t = timeit(lambda: ops.layernorm_bwd(dout, x, gam, mu, rs, dg, db, dres=dout), n=20)
print(f"layernorm bwd : {t:7.3f} ms  {4 * x.numel() * 2 / t / 1e9:6.2f} TB/s")
