"""Stand-alone reproduction of the attention forward's non-finite row found by tools/probe/nan_hunt_ref.py (gpurun_out/nan_hunt_block7_*.pt):
the (image, head) that fails as a B = 1, heads = 1, N = 65 problem, and variations that say what it takes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from unmore_amd import ops

import numpy as np
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden", "attn_wide_score_range_n65.npz"))
N, HD = 65, 64
x = torch.from_numpy(g["qkv_bf16_bits"]).view(torch.bfloat16).cuda()       # [N, 3, HD]


def run(xx, tag, n=N):
    xx = xx[:n].reshape(n, 3 * HD).contiguous()
    out, l = ops.attention_fwd(xx, 1, n, 1, need_lse=True)
    s = (xx[:, :HD].float() @ xx[:, HD:2 * HD].float().t()) * 0.125 * 1.4426950408889634
    ref = torch.softmax(s * 0.6931471805599453, dim=1) @ xx[:, 2 * HD:].float()
    nb = int((~torch.isfinite(out.float())).sum())
    err = float((out.float() - ref)[torch.isfinite(out.float())].abs().max()) if nb < out.numel() else float("nan")
    print(f"{tag:70s} non-finite out {nb:4d} lse {int((~torch.isfinite(l)).sum())}; score range of query 0: {float(s[0].min()):8.2f} .. {float(s[0].max()):8.2f} (argmax {int(s[0].argmax())}, argmin {int(s[0].argmin())}); "
          f"max err of the finite rows vs fp32 softmax {err:.3e}", flush=True)


run(x, "as found")
y = x.clone(); y[0, 0] *= 0.9
run(y, "query 0 scaled by 0.9 (range below 128)")
y = x.clone(); y[0, 0] *= 1.5
run(y, "query 0 scaled by 1.5")
y = x.clone(); y[64, 1] = 0
run(y, "key 64 zeroed")
run(x, "first 64 tokens only (one tile)", n=64)
y = x.clone(); amin = int(((x[0, 0].float() @ x[:, 1].float().t())).argmin()); y[amin, 1] = 0
run(y, f"the most negative key ({amin}) zeroed")
y = x.clone(); y[:, 2] = 1.0
run(y, "V = 1 everywhere")
# a synthetic case: one query, keys with scores +70 and -70 in the same tile
z = torch.zeros(65, 3, HD, device="cuda", dtype=torch.bfloat16)
z[:, 2] = torch.randn(65, HD, device="cuda").bfloat16()
z[0, 0, 0] = 8.0
z[3, 1, 0] = 70.0 / (8.0 * 0.125 * 1.4426950408889634)
z[9, 1, 0] = -70.0 / (8.0 * 0.125 * 1.4426950408889634)
run(z, "synthetic: scores +70 and -70 in tile 0, the rest 0")
z2 = z.clone(); z2[9, 1, 0] = 0; z2[64, 1, 0] = -70.0 / (8.0 * 0.125 * 1.4426950408889634)
run(z2, "synthetic: +70 in tile 0, -70 at key 64 (tile 1)")
z3 = z.clone(); z3[9, 1, 0] = 0; z3[3, 1, 0] = 0; z3[64, 1, 0] = 70.0 / (8.0 * 0.125 * 1.4426950408889634); z3[5, 1, 0] = -70.0 / (8.0 * 0.125 * 1.4426950408889634)
run(z3, "synthetic: -70 in tile 0, +70 at key 64 (tile 1)")

print("--- scan of the query's scale (one tile, N = 64)")
for f in (0.90, 0.91, 0.92, 0.93, 0.94, 0.95, 0.96, 0.97, 0.98, 0.99, 1.00):
    y = x[:64].clone(); y[0, 0] = (y[0, 0].float() * f).bfloat16()
    xx = y.reshape(64, 3 * HD).contiguous()
    out, l = ops.attention_fwd(xx, 1, 64, 1, need_lse=True)
    s = (xx[:, :HD].float() @ xx[:, HD:2 * HD].float().t()) * 0.125 * 1.4426950408889634
    srt = torch.sort(s[0], descending=True).values
    print(f"scale {f:.2f}: lse[0] {float(l[0]):10.4f} (fp32 reference {float(torch.logsumexp(s[0] * 0.6931471805599453, 0)):8.4f}) range {float(s[0].min()):8.2f} .. {float(s[0].max()):7.2f}; top-3 {srt[:3].tolist()}")
print("--- which single key, zeroed, removes the failure (N = 64)?")
s = (x[:64, 0].float() @ x[:64, 1].float().t()) * 0.125 * 1.4426950408889634
fix = []
for k in range(64):
    y = x[:64].clone(); y[k, 1] = 0
    out, l = ops.attention_fwd(y.reshape(64, 3 * HD).contiguous(), 1, 64, 1, need_lse=True)
    if bool(torch.isfinite(l[0])):
        fix.append((k, round(float(s[0, k]), 2)))
print("keys whose removal fixes it (key, its score):", fix)
print("scores of query 0 by key:", [round(v, 1) for v in s[0].tolist()])
