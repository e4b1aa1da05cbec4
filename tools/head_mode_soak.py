"""Round 6, review item 1(b): the boundary-distance head's collapsed training forward (the default since round 6) against the
four-convolution forward ('factored', the headline of rounds 2-5), same initial weights, same batch pool, bf16:

  1. first-step gradient: bf16 collapsed and bf16 factored, each against the fp32 FACTORED gradient of the same weights and batch
     (global cosine, worst per-tensor cosine, loss) -- the collapsed form must be at least as close as the factored one;
  2. a soak of N steps in each mode, losses printed side by side every 20 steps, 50-step averages compared.

    python tools/head_mode_soak.py [cfg2|ref] [steps] [--no-first-step]
cfg2 = dpt_base 384x384 batch 64 (BASELINE configs[1]); ref = the reference's own recipe (dpt_large, 128x128, batch 20)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from argparse import Namespace

import torch

from unmore_amd import synth
from unmore_amd.objectness_net import ObjectnessNet
from unmore_amd.trainer import TrainStep

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 800
BACKBONE, SIZE, BATCH = ("dpt_large", 128, 20) if cfg == "ref" else ("dpt_base", 384, 64)
dev = torch.device("cuda:0")
torch.manual_seed(0)
net0 = ObjectnessNet(dev, SIZE, BACKBONE, Namespace(use_bg_sdf=True, sdf_activation="tanh")).to(dev)
init = {k: v.detach().clone() for k, v in net0.state_dict().items()}
del net0
nb = 4
pool = []
for b in range(nb):
    _, cf, sdf, sal = synth.make_batch(BATCH, SIZE, SIZE, seed=100 + b)
    img = synth.blob_images(BATCH, SIZE, SIZE, seed=100 + b)
    pool.append(tuple(torch.from_numpy(a).to(dev) for a in (img, cf, sdf, sal)))


def make(dtype, mode, lr=1e-4):
    net = ObjectnessNet(dev, SIZE, BACKBONE, Namespace(use_bg_sdf=True, sdf_activation="tanh")).to(dev)
    net.load_state_dict(init, strict=True)
    net.set_compute_dtype(dtype)
    net.set_sdf_head_mode(mode)
    net.train()
    return net, TrainStep(net, lr=lr, lr_milestones=(10000, 20000), lr_gamma=0.1)


if "--no-first-step" not in sys.argv:
    grads, losses = {}, {}
    for name, dt, mode in (("fp32 factored", torch.float32, "factored"), ("bf16 factored", torch.bfloat16, "factored"),
                           ("bf16 collapsed (default)", torch.bfloat16, "auto")):
        net, st = make(dt, mode, lr=0.0)
        st.set_graph_mode("off")
        losses[name] = st.step(*pool[0]).cpu()
        grads[name] = {n: t.double().flatten().cpu() for n, t in st.G.items()}
        del net, st
        torch.cuda.empty_cache()
    ref = grads["fp32 factored"]
    cat = lambda g: torch.cat([g[n] for n in ref])
    cos = lambda a, b: float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-300))
    for name in ("bf16 factored", "bf16 collapsed (default)"):
        g = grads[name]
        worst = min(((cos(g[n], ref[n]), n) for n in ref if float(ref[n].norm()) > 0), key=lambda t: t[0])
        rel = max(float((g[n] - ref[n]).norm() / ref[n].norm()) for n in ref if float(ref[n].norm()) > 0)
        print(f"first step, {name:26s} vs fp32 factored: loss {losses[name][0].item():.5f} vs {losses['fp32 factored'][0].item():.5f}; "
              f"global gradient cosine {cos(cat(g), cat(ref)):.6f}; worst per-tensor cosine {worst[0]:.5f} ({worst[1]}); worst relative L2 {rel:.4f}", flush=True)
    del grads

hist = {}
rate = {}
torch.cuda.empty_cache()
torch.cuda.reset_peak_memory_stats()       # (the first-step leg above ran an fp32 step: not the soak's peak)
GRAPHS = os.environ.get("SOAK_GRAPHS", "auto")      # 'off': the eager two-stream schedule instead of the chain of per-stage graphs
for mode in ("auto", "factored"):
    net, st = make(torch.bfloat16, mode)
    st.set_graph_mode(GRAPHS)
    h = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(steps):
        h.append(st.step(*pool[it % nb]))
    torch.cuda.synchronize()
    rate[mode] = steps * BATCH / (time.perf_counter() - t0)
    hist[mode] = torch.stack(h).cpu()
    bad = (~torch.isfinite(hist[mode][:, 0])).nonzero().flatten()
    if bad.numel():
        b0 = int(bad[0])
        print(f"{mode}: NON-FINITE loss from step {b0 + 1} on; the ten steps before it: {[round(float(x), 4) for x in hist[mode][max(0, b0 - 10):b0 + 1, 0]]}", flush=True)
    print(f"{mode}: {rate[mode]:.1f} images/s sustained over {steps} steps incl. per-step host work; peak {torch.cuda.max_memory_allocated() / 2 ** 30:.1f} GiB", flush=True)
    del net, st
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
a, f = hist["auto"], hist["factored"]
print("step   total(collapsed) total(factored)   diff     | sdf(collapsed) sdf(factored)")
for it in range(19, steps, 20):
    print(f"{it + 1:5d}   {a[it, 0]:.4f}           {f[it, 0]:.4f}          {a[it, 0] - f[it, 0]:+.4f}  | {a[it, 2]:.4f}         {f[it, 2]:.4f}")
avg = lambda h: [h[i:i + 50, 0].mean().item() for i in range(0, steps - 49, 50)]
aa, af = avg(a), avg(f)
print("50-step averages, collapsed:", [round(v, 4) for v in aa])
print("50-step averages, factored: ", [round(v, 4) for v in af])
print(f"first 8 steps |diff|: {[round(abs(float(a[i, 0] - f[i, 0])), 5) for i in range(min(8, steps))]}")
print(f"largest |difference| of the 50-step averages: {max(abs(x - y) for x, y in zip(aa, af)):.4f}; final averages {aa[-1]:.4f} vs {af[-1]:.4f}; "
      f"loss falls in both: {aa[-1] < aa[0] and af[-1] < af[0]}")
assert torch.isfinite(a).all() and torch.isfinite(f).all(), "non-finite loss"
assert aa[-1] < aa[0] and af[-1] < af[0]
