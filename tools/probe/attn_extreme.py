import math, sys, torch
sys.path.insert(0, "/root/repo")
from unmore_amd import ops
dev = torch.device("cuda:0")
for N, off in ((65, -140.0), (65, -60.0), (65, 0.0), (200, -140.0), (200, 0.0)):
    HD, heads, B = 64, 2, 2
    gen = torch.Generator().manual_seed(N)
    x = torch.zeros(B, N, 3, heads, HD)
    x[:, :, 2] = torch.randn(B, N, heads, HD, generator=gen)
    x[:, :, 0, :, 1:] = 0.3 * torch.randn(B, N, heads, HD - 1, generator=gen)
    x[:, :, 1, :, 1:] = 0.3 * torch.randn(B, N, heads, HD - 1, generator=gen)
    x[:, :, 0, :, 0] = 8.0
    x[:, :, 1, :, 0] = off / (8.0 * 0.125 * 1.4426950408889634)
    xx = x.reshape(B * N, 3 * heads * HD).to(dev).bfloat16()
    g2 = torch.Generator().manual_seed(3)
    dout = torch.randn(B * N, heads * HD, generator=g2).to(dev).bfloat16()
    out, lse = ops.attention_fwd(xx, B, N, heads, need_lse=True)
    dqkv = ops.attention_bwd(xx, out, dout, lse, B, N, heads)
    xr = xx.double().view(B, N, 3, heads, HD).requires_grad_(True)
    q, k, v = xr[:, :, 0].transpose(1, 2), xr[:, :, 1].transpose(1, 2), xr[:, :, 2].transpose(1, 2)
    s = q @ k.transpose(-1, -2) / math.sqrt(HD)
    o = torch.softmax(s, dim=-1) @ v
    o.transpose(1, 2).reshape(B * N, heads * HD).backward(dout.double())
    ref = xr.grad.reshape(B * N, 3, heads, HD)
    got = dqkv.double().view(B * N, 3, heads, HD)
    oref = o.detach().transpose(1, 2).reshape(B * N, heads * HD)
    lref = torch.logsumexp(s.detach(), -1)
    rel = lambda a, b: float((a - b).norm() / b.norm())
    print(f"N {N} offset {off}: out rel {rel(out.double(), oref):.3e}; lse max err {float((lse.double().view(B, heads, N) - lref).abs().max()):.3e}; "
          f"dQ[1:] {rel(got[:, 0, :, 1:], ref[:, 0, :, 1:]):.3e} dK[1:] {rel(got[:, 1, :, 1:], ref[:, 1, :, 1:]):.3e} dV {rel(got[:, 2], ref[:, 2]):.3e}")
