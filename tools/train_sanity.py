"""Does the bf16 throughput path TRAIN at benchmark scale?  cfg2 model (ViT-B/16 DPT, 384x384, batch 64), the documented loss
flags, Adam 1e-4, N steps on a small pool of synthetic batches (images: structured blobs; labels: one ellipse each).  Prints the
five loss values and the allocator's current / peak bytes every 20 steps and checks that the total falls in 50-step averages (no
rise above 2 %) and stays finite.
    python tools/train_sanity.py [steps=200] [batches=4] [fp32_steps=0]
TRAIN_SANITY_DTYPE=fp32 trains in the fp32 parity mode instead of bf16.  TRAIN_SANITY_CFG=ref runs the reference's own recipe instead (dpt_large, 128 x 128, batch 20: the small-problem kernels, split-K).
With fp32_steps > 0 the first steps are repeated in fp32 parity mode from the same initial weights on the same batches and the two
loss trajectories are compared (bf16 storage must track fp32 within a few 1e-3 per step)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from argparse import Namespace

import torch

from unmore_amd import synth
from unmore_amd.objectness_net import ObjectnessNet
from unmore_amd.trainer import TrainStep

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n32 = int(sys.argv[3]) if len(sys.argv) > 3 else 0
BACKBONE, SIZE, BATCH = ("dpt_large", 128, 20) if os.environ.get("TRAIN_SANITY_CFG") == "ref" else ("dpt_base", 384, 64)
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = ObjectnessNet(dev, SIZE, BACKBONE, Namespace(use_bg_sdf=True, sdf_activation="tanh")).to(dev)
init = {k: v.detach().clone() for k, v in net.state_dict().items()}
DT = torch.float32 if os.environ.get("TRAIN_SANITY_DTYPE") == "fp32" else torch.bfloat16   # fp32: the parity mode on the bf16-plane kernels
net.set_compute_dtype(DT)
net.train()
step = TrainStep(net, lr=1e-4, lr_milestones=(10000, 20000), lr_gamma=0.1)
pool = []
for b in range(nb):
    _, cf, sdf, sal = synth.make_batch(BATCH, SIZE, SIZE, seed=100 + b)
    img = synth.blob_images(BATCH, SIZE, SIZE, seed=100 + b)
    pool.append(tuple(torch.from_numpy(a).to(dev) for a in (img, cf, sdf, sal)))
hist = []
t0 = time.perf_counter()
for it in range(steps):
    out5 = step.step(*pool[it % nb])
    hist.append(out5)
    if (it + 1) % 20 == 0:
        v = out5.cpu().tolist()
        print(f"step {it + 1:4d}  total {v[0]:.4f}  center {v[1]:.4f}  sdf {v[2]:.4f}  sdf-grad {v[3]:.4f}  bce {v[4]:.4f}  "
              f"| allocated {torch.cuda.memory_allocated() / 2 ** 30:.2f} GiB, peak {torch.cuda.max_memory_allocated() / 2 ** 30:.2f} GiB", flush=True)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
tot = torch.stack(hist)[:, 0].cpu()
assert torch.isfinite(tot).all(), "non-finite loss"
avg = [tot[i:i + 50].mean().item() for i in range(0, steps - 49, 50)]
print("50-step averages of the total loss:", [round(a, 4) for a in avg], f"| {steps * BATCH / dt:.1f} images/s incl. per-step host work")
assert avg[-1] < avg[0] and all(b < 1.02 * a for a, b in zip(avg, avg[1:])), "loss does not fall"

if n32 > 0:
    del step
    net32 = ObjectnessNet(dev, SIZE, BACKBONE, Namespace(use_bg_sdf=True, sdf_activation="tanh")).to(dev)
    net32.load_state_dict(init, strict=True)
    net32.set_compute_dtype(torch.float32)
    net32.train()
    step32 = TrainStep(net32, lr=1e-4, lr_milestones=(10000, 20000), lr_gamma=0.1)
    h32 = [step32.step(*pool[it % nb]) for it in range(n32)]
    t32 = torch.stack(h32)[:, 0].cpu()
    d = (tot[:n32] - t32).abs()
    print(f"first {n32} steps, bf16 vs fp32 total loss: max |diff| {d.max().item():.2e}; fp32 {[round(v, 4) for v in t32.tolist()]}")
    print(f"                                         bf16 {[round(v, 4) for v in tot[:n32].tolist()]}")
    if os.environ.get("TRAIN_SANITY_CFG") != "ref":
        assert d.max().item() < 2e-2      # cfg2 (64 images of 384 x 384 per step): the trajectories track each other
    else:
        # The reference recipe (ViT-L, 20 small images per step, loss rising over steps 2-4 before it falls) amplifies any
        # difference by ~1.5x per step (tools/probe/chaos_ref.py: two fp32-GRADE modes part the same way), so two precisions are
        # compared on ONE step from identical weights -- tests/test_parity_r2_gpu.py::test_bf16_vs_fp32_hip_at_the_other_workload_shapes
        # (loss within 2e-2, every gradient tensor cosine > 0.99, relative L2 < 0.12) -- and this trajectory difference is REPORTED:
        print(f"(reference recipe: reported, not asserted; step 1 differs by {d[0].item():.1e})")
        assert d[0].item() < 2e-3         # the first step is a single-step comparison: same weights, same batch
        if n32 >= 2:
            assert d[1].item() < 2e-3     # ... and the second is one amplification away from it (measured 1e-4)
