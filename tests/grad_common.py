"""Gradient parity on the HIP path's own linear piece (shared by the GPU gradient tests).

The HIP engine runs forward (saving activations), the fused loss kernel turns its maps into cotangents, the engine runs
backward; the float64 oracle then runs with the HIP path's ReLU decisions imposed at every site (oracle/mask_parity.py) and
is differentiated against the SAME cotangent maps.  What is left between the two is rounding only (measured 1e-6 .. 4e-6 of
max|g|), so the bar is 5e-5 -- no allowance for 'masks decided differently' or 'L1 signs decided differently' is needed."""
import torch

from oracle import mask_parity
from oracle import objectness_oracle as orc


def masked_gradient_check(net, sd, cfg_name, img, cf, sdf, sal, bar=5e-5, grads_out=None, masks_out=None, **head_kw):
    """net: unmore_amd ObjectnessNet on the GPU in fp32 mode holding `sd`.  Returns (worst max-norm error / max|g|, its
    parameter, worst relative L2, number of ReLU decisions that differ from float64's own)."""
    from unmore_amd import ops
    eng = net._engine()
    P = {n: p.detach() for n, p in net.named_parameters()}
    c_hip, s_hip, S = eng.forward(P, img.cuda(), save=True)
    masks = mask_parity.hip_relu_masks(S, (eng.center_layout, eng.sdf_layout))
    if masks_out is not None:
        masks_out.update(masks)      # the ReLU decisions this run took (callers compare two runs' decisions)
    out5, dpc, dps = ops.objectness_loss(c_hip, s_hip, cf.cuda(), sdf.cuda(), sal.cuda())
    nograd = net.nograd_names()
    G = {n: torch.zeros_like(P[n]) for n in P if n not in nograd}
    eng.backward(P, S, dpc, dps, G)
    torch.cuda.synchronize()
    sdo = {k: v.clone().double().requires_grad_(True) for k, v in sd.items()}
    out_o, flips = mask_parity.masked_forward(sdo, img.double(), orc.CONFIGS[cfg_name], masks, **head_kw)
    mask_parity.assert_flips_are_rounding(flips)
    for k, t in (("center_fields", c_hip), ("sdf_maps", s_hip)):
        assert (out_o[k].detach() - t.cpu().double()).abs().max().item() < 1e-4, k
    loss_o, _ = orc.loss_terms(out_o, cf.double(), sdf.double(), sal.double())
    assert abs(out5[0].item() - loss_o.item()) < 1e-4
    names = list(P)
    ref = torch.autograd.grad([out_o["center_fields"], out_o["sdf_maps"]], [sdo[n] for n in names],
                              grad_outputs=[dpc.cpu().double(), dps.cpu().double()], allow_unused=True)
    worst_inf, worst_n, worst_l2 = 0.0, "", 0.0
    for n, r in zip(names, ref):
        if n in nograd:
            assert r is None, n
            continue
        g = G[n].cpu().double()
        e_inf = (g - r).abs().max().item() / (r.abs().max().item() + 1e-300)
        e_l2 = (g - r).norm().item() / (r.norm().item() + 1e-300)
        if e_inf > worst_inf:
            worst_inf, worst_n = e_inf, n
        worst_l2 = max(worst_l2, e_l2)
        assert e_inf <= bar and e_l2 <= bar, (n, e_inf, e_l2)
    if grads_out is not None:
        grads_out.update(G)      # the engine-level gradients (callers compare them with what autograd delivered)
    return worst_inf, worst_n, worst_l2, sum(flips.values())
