"""Where does the first non-finite value of a reference-recipe soak come from?  Runs the recipe's TrainStep (dpt_large, 128x128, batch 20, bf16,
eager) up to the step before the first non-finite loss (found by a first pass), then repeats that step by hand -- forward, loss, backward --
and reports the first tensor with a non-finite element.    python tools/probe/nan_hunt_ref.py [mode=auto] [max_steps=2000]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from argparse import Namespace

import torch

from unmore_amd import ops, synth
from unmore_amd.objectness_net import ObjectnessNet
from unmore_amd.trainer import TrainStep

mode = sys.argv[1] if len(sys.argv) > 1 else "auto"
max_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
# NAN_HUNT_CFG = ref (default: dpt_large 128x128 batch 20) | cfg2 (dpt_base 384x384 batch 64) | cfg4 (dpt_large14 518x518 batch 16);
# NAN_HUNT_DTYPE = bf16 (default) | fp32
BACKBONE, SIZE, BATCH = {"ref": ("dpt_large", 128, 20), "cfg2": ("dpt_base", 384, 64), "cfg4": ("dpt_large14", 518, 16)}[os.environ.get("NAN_HUNT_CFG", "ref")]
DT = torch.float32 if os.environ.get("NAN_HUNT_DTYPE") == "fp32" else torch.bfloat16
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = ObjectnessNet(dev, SIZE, BACKBONE, Namespace(use_bg_sdf=True, sdf_activation="tanh")).to(dev)
net.set_compute_dtype(DT)
net.set_sdf_head_mode(mode)
net.train()
pool = []
for b in range(4):
    _, cf, sdf, sal = synth.make_batch(BATCH, SIZE, SIZE, seed=100 + b)
    img = synth.blob_images(BATCH, SIZE, SIZE, seed=100 + b)
    pool.append(tuple(torch.from_numpy(a).to(dev) for a in (img, cf, sdf, sal)))
st = TrainStep(net, lr=1e-4, lr_milestones=(10000, 20000), lr_gamma=0.1).set_graph_mode("off")


def fin(t):
    return bool(torch.isfinite(t.float()).all())


def stats(name, t):
    t = t.float()
    bad = int((~torch.isfinite(t)).sum())
    print(f"    {name:60s} shape {tuple(t.shape)} non-finite {bad} max|finite| {float(t[torch.isfinite(t)].abs().max()) if bad < t.numel() else float('nan'):.4e}", flush=True)


it = 0
hist = []
while it < max_steps:
    snap = None
    if it % 50 == 0:
        torch.cuda.synchronize()
    # keep a copy of the optimizer state every step is too slow: check the loss every step (one sync), snapshot lazily
    p0, m0, v0 = st.flat_p.clone(), st.m.clone(), st.v.clone()
    out5 = st.step(*pool[it % 4])
    hist.append(out5)
    if (it + 1) % 250 == 0:
        print(f"step {it + 1}: loss {[round(v, 4) for v in out5.tolist()]}", flush=True)
    if fin(out5) and not fin(st.flat_g):
        # the loss (taken in the forward pass) is finite but the step's BACKWARD produced non-finite gradients
        print(f"step {it + 1}: finite loss {out5.tolist()} but non-finite gradients; the 12 losses before: {[round(float(h[0]), 4) for h in hist[-13:-1]]}", flush=True)
        st.flat_p.copy_(p0); st.m.copy_(m0); st.v.copy_(v0)
        eng = net._engine()
        eng.cache.clear()
        P = {n: p.detach() for n, p in net.named_parameters()}
        img, cf, sdf, sal = pool[it % 4]
        c, s, S = eng.forward(P, img, save=True)
        stats("center_fields", c)
        stats("sdf_maps", s)
        print(f"    sdf_maps: share of |out| == 1 exactly: {float((s.abs() == 1).float().mean()):.4f}; max |out| {float(s.abs().max()):.8f}")
        for k, v in S["heads"][1].items():
            if torch.is_tensor(v):
                stats("sdf head saved: " + k, v)
        l5, dpc, dps = ops.objectness_loss(c, s, cf, sdf, sal)
        stats("d loss / d center", dpc)
        stats("d loss / d sdf", dps)
        G = {n: torch.zeros_like(P[n]) for n in P if n not in net.nograd_names()}
        seen = set()

        def cb(stage, wg):
            torch.cuda.synchronize()
            bad = [n for n, g in G.items() if n not in seen and not fin(g)]
            seen.update(bad)
            if bad:
                print(f"  after backward stage '{stage}': {len(bad)} newly non-finite gradient tensors {bad[:6]}", flush=True)
            for n in bad[:4]:
                stats(n, G[n])
        real_bwd = ops.attention_bwd
        state = {"saved": False, "call": 0}

        def spy(qkv, out, dout, lse, B_, N_, heads_):
            dq = real_bwd(qkv, out, dout, lse, B_, N_, heads_)
            state["call"] += 1
            if not state["saved"] and not fin(dq):
                state["saved"] = True
                bad = ~torch.isfinite(dq.float())
                rows = bad.any(1).nonzero().flatten()
                cols = bad.any(0).nonzero().flatten()
                print(f"  attention_bwd call {state['call']} (block {max(net.cfg['hooks']) + 1 - state['call']}): inputs finite qkv {fin(qkv)} out {fin(out)} dout {fin(dout)} lse {fin(lse)}; "
                      f"dqkv non-finite {int(bad.sum())} in rows {rows[:6].tolist()}..({rows.numel()}) cols {cols[:4].tolist()}..{cols[-2:].tolist()} ({cols.numel()})", flush=True)
                out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "gpurun_out")
                torch.save({"qkv": qkv.cpu(), "out": out.cpu(), "dout": dout.cpu(), "lse": lse.cpu(), "dqkv": dq.cpu()}, os.path.join(out_dir, f"nan_hunt_attn_bwd_{mode}.pt"))
            return dq
        ops.attention_bwd = spy
        eng.backward(P, S, dpc, dps, G, stage_cb=cb)
        ops.attention_bwd = real_bwd
        break
    if not fin(out5):
        print(f"step {it + 1}: loss {out5.tolist()} -- repeating it by hand from the state before it", flush=True)
        st.flat_p.copy_(p0); st.m.copy_(m0); st.v.copy_(v0)
        eng = net._engine()
        eng.cache.clear()
        P = {n: p.detach() for n, p in net.named_parameters()}
        print("  parameters finite:", all(fin(p) for p in P.values()), "| Adam m, v finite:", fin(m0), fin(v0))
        img, cf, sdf, sal = pool[it % 4]
        c, s, S = eng.forward(P, img, save=True)
        print("  forward:")
        stats("center_fields", c)
        stats("sdf_maps", s)
        for k, v in S["heads"][1].items():
            if torch.is_tensor(v):
                stats("sdf head saved: " + k, v)
        if "path" in S and S["path"] is not None:
            stats("path (small feature map)", S["path"])
        # the transformer: first saved tensor with a non-finite element, block by block in execution order
        order = ("x", "mean1", "rstd1", "ln1", "qkv", "lse", "att", "x1", "mean2", "rstd2", "ln2", "hpre", "h")
        Nt = S["gh"] * S["gw"] + 1
        found = False
        for i, bs in enumerate(S["blocks"]):
            if bs is None:
                continue
            for k in order:
                t = bs.get(k)
                if torch.is_tensor(t) and not fin(t):
                    tf = t.float()
                    bad = ~torch.isfinite(tf)
                    rows = bad.reshape(bad.shape[0], -1).any(1).nonzero().flatten() if bad.dim() >= 2 else bad.nonzero().flatten()
                    print(f"  FIRST non-finite tensor of the transformer: block {i} '{k}' shape {tuple(t.shape)}: {int(bad.sum())} elements in {rows.numel()} rows; "
                          f"first rows {rows[:8].tolist()} (token rows per image: {Nt}; image of the first row: {int(rows[0]) // Nt if t.shape[0] % Nt == 0 else '?'})")
                    if bad.dim() == 2:
                        r0 = int(rows[0])
                        print(f"    columns of row {r0} that are non-finite: {bad[r0].nonzero().flatten()[:16].tolist()} ... ({int(bad[r0].sum())} of {bad.shape[1]})")
                        vals = tf[r0][bad[r0]][:8].tolist()
                        print(f"    their values: {vals}")
                    # what went INTO it: the largest magnitudes of the block's earlier tensors
                    for k2 in order:
                        t2 = bs.get(k2)
                        if torch.is_tensor(t2):
                            stats(f"block {i} {k2}", t2)
                        if k2 == k:
                            break
                    found = True
                    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "gpurun_out")
                    os.makedirs(out_dir, exist_ok=True)
                    torch.save({kk: (vv.cpu() if torch.is_tensor(vv) else vv) for kk, vv in bs.items() if kk in ("qkv", "lse", "att")},
                               os.path.join(out_dir, f"nan_hunt_block{i}_{mode}.pt"))
                    # does the kernel reproduce it stand-alone on the saved operand?
                    att2, lse2 = ops.attention_fwd(bs["qkv"], S["B"], Nt, net.cfg["heads"], need_lse=True)
                    print(f"    stand-alone attention_fwd on the saved qkv: lse non-finite {int((~torch.isfinite(lse2)).sum())}, out non-finite {int((~torch.isfinite(att2.float())).sum())}")
                    break
            if found:
                break
        l5, dpc, dps = ops.objectness_loss(c, s, cf, sdf, sal)
        print("  loss terms:", l5.tolist())
        stats("d loss / d center", dpc)
        stats("d loss / d sdf", dps)
        G = {n: torch.zeros_like(P[n]) for n in P if n not in net.nograd_names()}
        eng.backward(P, S, dpc, dps, G)
        torch.cuda.synchronize()
        print("  gradients with non-finite elements:")
        nbad = 0
        for n, g in G.items():
            if not fin(g):
                nbad += 1
                if nbad <= 12:
                    stats(n, g)
        print(f"  {nbad} of {len(G)} gradient tensors")
        # the previous step's gradients and update
        break
    it += 1
else:
    print(f"no non-finite loss or gradient in {max_steps} steps ({mode}, {os.environ.get('NAN_HUNT_CFG', 'ref')}, {'fp32' if DT == torch.float32 else 'bf16'})")
