"""Every kernel of one transformer block, forward and backward, at the cfg2 ViT-B shape (B=64, N=577, D=768,
12 heads) -- the "attention block" the north star names.  Prints time, TFLOP/s (or TB/s) and % of the 2.5 PFLOP/s
dense bf16 peak per kernel, then the block total.  Run on the MI355X:
    python tools/vit_block_bench.py [D] [heads] [B] [N]
UMR_GEMM_TILE=128|256 forces the NT/TN tile size (A/B)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from unmore_amd import ops, _lib as L
from kbench import timeit

D = int(sys.argv[1]) if len(sys.argv) > 1 else 768
H = int(sys.argv[2]) if len(sys.argv) > 2 else 12
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
N = int(sys.argv[4]) if len(sys.argv) > 4 else 577
M = B * N
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
PEAK = 2500.0


def rnd(*shape, scale=1.0, dt=torch.bfloat16):
    return (torch.randn(shape, generator=g) * scale).to(dev).to(dt)


x = rnd(M, D)
x4 = rnd(M, 4 * D)
x3 = rnd(M, 3 * D, scale=0.5)
w_qkv, w_proj, w_fc1, w_fc2 = rnd(3 * D, D, scale=0.03), rnd(D, D, scale=0.03), rnd(4 * D, D, scale=0.03), rnd(D, 4 * D, scale=0.03)
w_qkv_t, w_fc1_t, w_fc2_t = w_qkv.t().contiguous(), w_fc1.t().contiguous(), w_fc2.t().contiguous()
b3, b1, b4 = torch.zeros(3 * D, device=dev), torch.zeros(D, device=dev), torch.zeros(4 * D, device=dev)
gam, bet = torch.ones(D, device=dev), torch.zeros(D, device=dev)
o1, o3, o4, o4b = torch.empty_like(x), torch.empty_like(x3), torch.empty_like(x4), torch.empty_like(x4)
dW = {k: torch.zeros(s, device=dev) for k, s in (("qkv", (3 * D, D)), ("proj", (D, D)), ("fc1", (4 * D, D)), ("fc2", (D, 4 * D)))}
db = {k: torch.zeros(s, device=dev) for k, s in (("qkv", 3 * D), ("proj", D), ("fc1", 4 * D), ("fc2", D))}
att, lse = ops.attention_fwd(x3, B, N, H)
_, mu, rs = ops.layernorm_fwd(x, gam, bet)
dg, dbt = torch.zeros(D, device=dev), torch.zeros(D, device=dev)

rows = []


def gemm(name, fn, m, n, k):
    t = timeit(fn, n=10)
    tf = 2.0 * m * n * k / t / 1e9
    rows.append((name, t, f"{tf:7.1f} TFLOP/s {100 * tf / PEAK:5.1f} %", 2.0 * m * n * k))


def mem(name, fn, nbytes):
    t = timeit(fn, n=10)
    rows.append((name, t, f"{nbytes / t / 1e9:7.2f} TB/s", 0.0))


fa = 4.0 * B * H * N * N * (D // H)
# ---- forward
mem("fwd ln1", lambda: ops.layernorm_fwd(x, gam, bet), 2 * x.numel() * 2)
gemm("fwd qkv", lambda: ops.gemm_nt(x, w_qkv, b3, out=o3), M, 3 * D, D)
t = timeit(lambda: ops.attention_fwd(x3, B, N, H), n=10)
rows.append(("fwd attention", t, f"{fa / t / 1e9:7.1f} TFLOP/s {100 * fa / t / 1e9 / PEAK:5.1f} %", fa))
gemm("fwd proj (+res)", lambda: ops.gemm_nt(x, w_proj, b1, aux=x, out=o1), M, D, D)
mem("fwd ln2", lambda: ops.layernorm_fwd(x, gam, bet), 2 * x.numel() * 2)
gemm("fwd fc1 (GELU, save pre)", lambda: ops.gemm_nt(x, w_fc1, b4, act=L.ACT_GELU, c2_mode=2, out=o4, out2=o4b), M, 4 * D, D)
gemm("fwd fc2 (+res)", lambda: ops.gemm_nt(x4, w_fc2, b1, aux=x, out=o1), M, D, 4 * D)
n_fwd = len(rows)
# ---- backward
gemm("bwd fc2 wgrad", lambda: ops.gemm_tn(x, x4, dW=dW["fc2"], dbias=db["fc2"]), M, D, 4 * D)
gemm("bwd fc2 dgrad (dGELU)", lambda: ops.gemm_nt(x, w_fc2_t, None, aux=x4, mask_dgelu=True, out=o4), M, 4 * D, D)
gemm("bwd fc1 wgrad", lambda: ops.gemm_tn(x4, x, dW=dW["fc1"], dbias=db["fc1"]), M, 4 * D, D)
gemm("bwd fc1 dgrad", lambda: ops.gemm_nt(x4, w_fc1_t, None, out=o1), M, D, 4 * D)
mem("bwd ln2", lambda: ops.layernorm_bwd(x, x, gam, mu, rs, dg, dbt, dres=x), 4 * x.numel() * 2)
gemm("bwd proj wgrad", lambda: ops.gemm_tn(x, x, dW=dW["proj"], dbias=db["proj"]), M, D, D)
gemm("bwd proj dgrad", lambda: ops.gemm_nt(x, w_proj, None, out=o1), M, D, D)
t = timeit(lambda: ops.attention_bwd(x3, att, x, lse, B, N, H), n=10)
rows.append(("bwd attention", t, f"{2.5 * fa / t / 1e9:7.1f} TFLOP/s {100 * 2.5 * fa / t / 1e9 / PEAK:5.1f} %", 2.5 * fa))
gemm("bwd qkv wgrad", lambda: ops.gemm_tn(x3, x, dW=dW["qkv"], dbias=db["qkv"]), M, 3 * D, D)
gemm("bwd qkv dgrad", lambda: ops.gemm_nt(x3, w_qkv_t, None, out=o1), M, D, 3 * D)
mem("bwd ln1", lambda: ops.layernorm_bwd(x, x, gam, mu, rs, dg, dbt, dres=x), 4 * x.numel() * 2)

print(f"ViT block at M={M} tokens (B={B}, N={N}), D={D}, {H} heads, tile override {os.environ.get('UMR_GEMM_TILE', '-')}")
for i, (name, t, s, _) in enumerate(rows):
    if i == n_fwd:
        print("  --")
    print(f"  {name:28s} {t * 1e3:8.1f} us  {s}")
tot_f, fl_f = sum(r[1] for r in rows[:n_fwd]), sum(r[3] for r in rows[:n_fwd])
tot_b, fl_b = sum(r[1] for r in rows[n_fwd:]), sum(r[3] for r in rows[n_fwd:])
print(f"  forward  {tot_f:7.3f} ms  {fl_f / tot_f / 1e9:7.1f} TFLOP/s ({100 * fl_f / tot_f / 1e9 / PEAK:4.1f} % of peak)")
print(f"  backward {tot_b:7.3f} ms  {fl_b / tot_b / 1e9:7.1f} TFLOP/s ({100 * fl_b / tot_b / 1e9 / PEAK:4.1f} % of peak)")
print(f"  block    {tot_f + tot_b:7.3f} ms  {(fl_f + fl_b) / (tot_f + tot_b) / 1e9:7.1f} TFLOP/s")
