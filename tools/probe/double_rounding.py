"""What the second bf16 rounding in the large-tile epilogues costs.  The persistent 256x256 kernel applies a
residual add / mask to the value already rounded to bf16 -- bf16(bf16(acc + bias) + aux) -- while the 128x128 kernel adds in f32 and
rounds once.  UMR_GEMM_TILE=128 forces the single-rounding kernel everywhere: one bf16 train step in each mode against the fp32
step of the same engine, same weights and batch (dpt_base, 384x384, B = 4: 2308 tokens, every large-tile path active)."""
import os
import sys
from argparse import Namespace

import torch

sys.path.insert(0, ".")
from unmore_amd import synth  # noqa: E402
from unmore_amd.hashrng import hash_init  # noqa: E402
from unmore_amd.objectness_net import ObjectnessNet  # noqa: E402
from unmore_amd.trainer import TrainStep  # noqa: E402
from unmore_amd import ops  # noqa: E402

B, H, W = 4, 384, 384
_, cf, sdf, sal = (torch.from_numpy(a).cuda() for a in synth.make_batch(B, H, W, seed=9))
img = torch.from_numpy(synth.blob_images(B, H, W, seed=9)).cuda()


def step(dtype, tile):
    if tile:
        ops.set_debug_option("UMR_GEMM_TILE", tile)
    else:
        ops.set_debug_option("UMR_GEMM_TILE", None)
    net = ObjectnessNet("cuda:0", H, "dpt_base", Namespace(use_bg_sdf=True, sdf_activation="tanh"))
    net.load_state_dict({k: torch.from_numpy(hash_init(k, tuple(v.shape), "base")) for k, v in net.state_dict().items()}, strict=True)
    net = net.to("cuda:0")
    net.set_compute_dtype(dtype)
    st = TrainStep(net, lr=0.0).set_graph_mode("off")
    loss = st.step(img, cf, sdf, sal)[0].item()
    g = {n: t.clone() for n, t in st.G.items()}
    ops.set_debug_option("UMR_GEMM_TILE", None)
    return loss, g


l32, g32 = step(torch.float32, None)
ref = torch.cat([g32[n].flatten() for n in g32]).double()
for name, tile in (("double rounding (256x256 kernels, default)", None), ("single rounding (128x128 kernels everywhere)", "128")):
    l, g = step(torch.bfloat16, tile)
    a = torch.cat([g[n].flatten() for n in g32]).double()
    cos = (torch.dot(a, ref) / (a.norm() * ref.norm())).item()
    rel = ((a - ref).norm() / ref.norm()).item()
    worst = max(((g[n].double() - g32[n].double()).norm() / (g32[n].double().norm() + 1e-300)).item() for n in g32 if g32[n].norm() > 0)
    print(f"{name:48s} loss {l:.6f} (fp32 {l32:.6f}, diff {abs(l - l32):.2e})  global cosine {cos:.6f}  global rel-L2 {rel:.4f}  worst per-tensor rel-L2 {worst:.4f}")
