"""Where does smoke()'s fp32 gradient disagreement come from?  (7.6e-4 in round 1 -> 9.15e-3 in round 2 on
the same model, inputs and CPU oracle.)  One process, the dpt_tiny smoke case, every combination of the two A/B switches that
changed between the rounds -- fp32 product mode (exact f32 MFMA | three-way bf16 split) and the boundary-distance head's backward
(layer-by-layer GEMMs | algebraic) -- against three references: the CPU fp32 oracle (what smoke compared with), the float64
oracle, and the float64 oracle with the HIP path's own ReLU decisions imposed (oracle/mask_parity.py).  Prints, per
combination, the worst parameter (max-norm error / max|g|) under each reference.

    python tools/smoke_ab.py            (on the GPU box)"""
import os
import sys
from argparse import Namespace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import mask_parity  # noqa: E402
from oracle import objectness_oracle as orc  # noqa: E402
from unmore_amd import ops, synth  # noqa: E402
from unmore_amd.hashrng import hash_init  # noqa: E402
from unmore_amd.objectness_net import ObjectnessNet  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    cfg = orc.CONFIGS["dpt_tiny"]
    B, H, W = 2, 64, 64
    img, cf, sdf, sal = (torch.from_numpy(a) for a in synth.make_batch(B, H, W, seed=1))
    spec = orc.state_dict_spec(cfg)
    sd = {k: torch.from_numpy(hash_init(k, s, "tiny")) for k, s in spec.items()}

    def oracle_grads(dtype, masks=None):
        sdo = {k: v.clone().to(dtype).requires_grad_(True) for k, v in sd.items()}
        if masks is None:
            out = orc.forward(sdo, img.to(dtype), cfg)
        else:
            out, _ = mask_parity.masked_forward(sdo, img.to(dtype), cfg, masks)
        loss, _ = orc.loss_terms(out, cf.to(dtype), sdf.to(dtype), sal.to(dtype))
        loss.backward()
        return {k: v.grad for k, v in sdo.items()}

    g32, g64 = oracle_grads(torch.float32), oracle_grads(torch.float64)
    for f32_mode in ("exact", "x3"):
        for head_bwd in ("gemm", "algebraic"):
            ops.set_f32_mode(f32_mode)
            net = ObjectnessNet(dev, 64, "dpt_tiny", Namespace(use_bg_sdf=True, sdf_activation="tanh"))
            net.load_state_dict(sd, strict=True)
            net = net.to(dev)
            net.set_linear_head_backward(head_bwd)
            eng = net._engine()
            P = {n: p.detach() for n, p in net.named_parameters()}
            c, s, S = eng.forward(P, img.to(dev), save=True)
            masks = mask_parity.hip_relu_masks(S, (eng.center_layout, eng.sdf_layout))
            _, dpc, dps = ops.objectness_loss(c, s, cf.to(dev), sdf.to(dev), sal.to(dev))
            nograd = net.nograd_names()
            G = {n: torch.zeros_like(P[n]) for n in P if n not in nograd}
            eng.backward(P, S, dpc, dps, G)
            torch.cuda.synchronize()
            g64m = oracle_grads(torch.float64, masks)
            line = []
            for tag, ref in (("cpu fp32", g32), ("float64", g64), ("float64 + HIP masks", g64m)):
                worst, wn = 0.0, ""
                for n, g in G.items():
                    r = ref[n].double()
                    e = ((g.cpu().double() - r).abs().max() / (r.abs().max() + 1e-300)).item()
                    if e > worst:
                        worst, wn = e, n
                line.append(f"{tag}: {worst:.2e} ({wn})")
            print(f"f32 products {f32_mode:5s} | sdf-head backward {head_bwd:9s} | " + " | ".join(line), flush=True)
    # the yardstick: the two CPU oracles against each other
    worst, wn = 0.0, ""
    for n, r in g64.items():
        if r is None:
            continue
        e = ((g32[n].double() - r).abs().max() / (r.abs().max() + 1e-300)).item()
        if e > worst:
            worst, wn = e, n
    print(f"CPU fp32 oracle vs float64 oracle: {worst:.2e} ({wn})")


if __name__ == "__main__":
    main()
