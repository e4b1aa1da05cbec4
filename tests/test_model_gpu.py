"""End-to-end parity of the HIP ObjectnessNet against the committed golden fixtures
(made by the reference's own modules) and against the CPU oracle (forward + gradients).
fp32 mode carries the 1e-4 contract; bf16 mode is checked at a documented looser bound."""
import os
from argparse import Namespace

import numpy as np
import pytest
import torch

from oracle import objectness_oracle as orc
from unmore_amd.hashrng import hash_init, uniform01

pytestmark = pytest.mark.gpu
ARGS = Namespace(use_bg_sdf=True, sdf_activation="tanh")


def _net(backbone, tag, dtype=torch.float32, args=ARGS):
    from unmore_amd.objectness_net import ObjectnessNet
    assert torch.cuda.is_available()
    net = ObjectnessNet("cuda:0", 128, backbone, args)
    sd = {k: torch.from_numpy(hash_init(k, tuple(v.shape), tag)) for k, v in net.state_dict().items()}
    net.load_state_dict(sd, strict=True)
    net = net.to("cuda:0").to(torch.float32)
    net.set_compute_dtype(dtype)
    return net, sd


@pytest.mark.parametrize("cfg,tag,img,fname,B,H,W", [
    ("dpt_tiny", "tiny", "tiny64x64", "fwd_dpt_tiny_64x64.npz", 2, 64, 64),
    ("dpt_tiny", "tiny", "tiny96x64", "fwd_dpt_tiny_96x64.npz", 2, 96, 64),
    ("dpt_base", "base", "base128", "fwd_dpt_base_128.npz", 1, 128, 128),
    ("dpt_large", "large", "large128", "fwd_dpt_large_128.npz", 1, 128, 128),
])
def test_forward_fp32_matches_reference_golden(golden_dir, cfg, tag, img, fname, B, H, W):
    g = np.load(os.path.join(golden_dir, fname))
    net, _ = _net(cfg, tag)
    net.eval()
    x = torch.from_numpy(uniform01(f"img:{img}", (B, 3, H, W))).cuda()
    with torch.no_grad():
        out = net(images=x)
        out2 = net.get_prediction(x)
    assert out["center_fields"].shape == (B, 2, H, W) and out["sdf_maps"].shape == (B, 1, H, W)
    assert out["center_fields"].dtype == torch.float32 and out["center_fields"].is_cuda
    # the north-star contract: within 1e-4 (fp32) of the reference CPU path
    np.testing.assert_allclose(out["center_fields"].cpu().numpy(), g["center_fields"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(out["sdf_maps"].cpu().numpy(), g["sdf_maps"], atol=1e-4, rtol=0)
    assert torch.equal(out["center_fields"], out2["center_fields"]) and torch.equal(out["sdf_maps"], out2["sdf_maps"])


@pytest.mark.parametrize("cfg,tag,img,fname,B,H,W", [
    ("dpt_tiny", "tiny", "tiny96x64", "fwd_dpt_tiny_96x64.npz", 2, 96, 64),
    ("dpt_base", "base", "base128", "fwd_dpt_base_128.npz", 1, 128, 128),
])
def test_forward_bf16_close_to_reference_golden(golden_dir, cfg, tag, img, fname, B, H, W):
    """bf16 storage / fp32 accumulate cannot meet 1e-4; bound: 3e-2 abs on O(1) fields (documented in DESIGN.md)."""
    g = np.load(os.path.join(golden_dir, fname))
    net, _ = _net(cfg, tag, torch.bfloat16)
    net.eval()
    x = torch.from_numpy(uniform01(f"img:{img}", (B, 3, H, W))).cuda()
    with torch.no_grad():
        out = net(images=x)
    np.testing.assert_allclose(out["center_fields"].cpu().numpy(), g["center_fields"], atol=3e-2, rtol=0)
    np.testing.assert_allclose(out["sdf_maps"].cpu().numpy(), g["sdf_maps"], atol=3e-2, rtol=0)


def _labels(B, H, W, seed):
    rng = np.random.RandomState(seed)
    masks = np.zeros((B, H, W), np.uint8)
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    for b in range(B):
        cy, cx = rng.uniform(0.25, 0.75) * H, rng.uniform(0.25, 0.75) * W
        ry, rx = rng.uniform(H / 8, H / 3), rng.uniform(W / 8, W / 3)
        masks[b] = (((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2) <= 1
    cf, sdf, sal = orc.synth_labels(masks)
    return torch.from_numpy(cf), torch.from_numpy(sdf).unsqueeze(1), torch.from_numpy(sal).unsqueeze(1)


@pytest.mark.parametrize("H,W", [(64, 64), (96, 64)])
def test_backward_fp32_matches_oracle_autograd(H, W):
    """Every parameter gradient of the 4-term loss, delivered through the autograd.Function boundary (net(images) ->
    objectness_loss -> loss.backward() -> p.grad).  Three statements:
      (1) p.grad is bit-identical to what the engine-level forward / loss kernel / backward produce (same kernels, another door);
      (2) those gradients agree with the float64 oracle to 5e-5 * max|g| on the HIP path's own linear piece (the oracle runs with
          the HIP path's ReLU decisions imposed, tests/grad_common.py) -- the statement that is about arithmetic;
      (3) against the oracle's OWN ReLU decisions the whole-gradient relative L2 error stays below 5e-4 (a decision that falls
          the other way within rounding of zero moves single entries by ~1e-3 * max|g| -- which one does depends on the summation
          order, e.g. on split-K -- so the max-norm is printed, not asserted; round 2 asserted it and the bar followed the luck)."""
    from unmore_amd.loss import objectness_loss
    from grad_common import masked_gradient_check
    B = 2
    net, sd = _net("dpt_tiny", "tiny")
    net.train()
    x = torch.from_numpy(uniform01(f"img:tiny{H}x{W}", (B, 3, H, W)))
    gc, gs, sal = _labels(B, H, W, 0)
    grads = {}
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        sdo = {k: v.clone().to(dt).requires_grad_(True) for k, v in sd.items()}
        out_o = orc.forward(sdo, x.to(dt), orc.CONFIGS["dpt_tiny"])
        loss_o, terms_o = orc.loss_terms(out_o, gc.to(dt), gs.to(dt), sal.to(dt))
        loss_o.backward()
        grads[tag] = {k: v.grad for k, v in sdo.items()}
    # HIP path through the autograd.Function boundary
    out = net(images=x.cuda())
    loss = objectness_loss(out, gc.cuda(), gs.cuda(), sal.cuda())
    loss.backward()
    assert abs(loss.item() - loss_o.item()) < 1e-4
    nograd = net.nograd_names()
    autograd_grads = {n: (p.grad.clone() if p.grad is not None else None) for n, p in net.named_parameters()}
    G = {}
    worst_inf, worst_n, worst_l2, flips = masked_gradient_check(net, sd, "dpt_tiny", x, gc, gs, sal, grads_out=G)     # (2)
    num = den = num_c = 0.0
    worst_mine, worst_cpu = 0.0, 0.0
    for n, p in net.named_parameters():
        ref = grads["f64"][n]
        if n in nograd:
            assert autograd_grads[n] is None and ref is None, n
            continue
        g = autograd_grads[n]
        assert g is not None and g.shape == p.shape, n
        assert torch.equal(g, G[n]), f"{n}: autograd delivered something else than the engine computes"                 # (1)
        scale = ref.abs().max().item() + 1e-12
        worst_mine = max(worst_mine, (g.cpu().double() - ref).abs().max().item() / scale)
        worst_cpu = max(worst_cpu, (grads["f32"][n].double() - ref).abs().max().item() / scale)
        num += (g.cpu().double() - ref).pow(2).sum().item()
        num_c += (grads["f32"][n].double() - ref).pow(2).sum().item()
        den += ref.pow(2).sum().item()
    l2, l2_c = (num / den) ** 0.5, (num_c / den) ** 0.5
    print(f"linear piece: worst {worst_inf:.2e} * max|g| ({worst_n}), {flips} ReLU decisions differ from float64's own; against the "
          f"unmasked oracle: relative L2 {l2:.2e} (reference-style CPU fp32 {l2_c:.2e}), worst entry {worst_mine:.2e} (CPU fp32 {worst_cpu:.2e})")
    assert l2 <= 5e-4                                                                                                  # (3)


def test_backward_bf16_gradients_are_close():
    from unmore_amd.loss import objectness_loss
    B, H, W = 2, 64, 64
    net, sd = _net("dpt_tiny", "tiny", torch.bfloat16)
    net.train()
    x = torch.from_numpy(uniform01("img:tiny64x64", (B, 3, H, W)))
    gc, gs, sal = _labels(B, H, W, 0)
    sdo = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    loss_o, _ = orc.loss_terms(orc.forward(sdo, x, orc.CONFIGS["dpt_tiny"]), gc, gs, sal)
    loss_o.backward()
    loss = objectness_loss(net(images=x.cuda()), gc.cuda(), gs.cuda(), sal.cuda())
    loss.backward()
    assert abs(loss.item() - loss_o.item()) < 2e-2
    # cosine similarity of the full gradient vector
    a = torch.cat([p.grad.flatten().cpu() for n, p in net.named_parameters() if p.grad is not None])
    b = torch.cat([sdo[n].grad.flatten() for n, p in net.named_parameters() if p.grad is not None])
    cos = torch.dot(a, b) / (a.norm() * b.norm())
    assert cos > 0.99, cos


@pytest.mark.parametrize("bg,act", [(True, "sine"), (True, None), (True, "relu"), (False, "tanh")])
def test_backward_fp32_other_sdf_activations(bg, act):
    """objectness_net.py:119-164: the 'sine' (SinActivation, :30-35), activation-free, ReLU and use_bg_sdf=False variants of the boundary-distance head,
    forward and every parameter gradient vs the oracle's float64 autograd.  (sin is not invertible from its value: the
    backward pass is fed the pre-activation.)"""
    from unmore_amd.loss import objectness_loss
    B, H, W = 2, 64, 64
    args = Namespace(use_bg_sdf=bg, sdf_activation=act)
    net, sd = _net("dpt_tiny", "tiny", args=args)
    net.train()
    x = torch.from_numpy(uniform01("img:tiny64x64", (B, 3, H, W)))
    gc, gs, sal = _labels(B, H, W, 0)
    sdo = {k: v.clone().double().requires_grad_(True) for k, v in sd.items()}
    out_o = orc.forward(sdo, x.double(), orc.CONFIGS["dpt_tiny"], bg, act)
    # the binary-mask term takes log(sigmoid(.)) of the map: fine for every variant; all four terms as documented
    loss_o, _ = orc.loss_terms(out_o, gc.double(), gs.double(), sal.double())
    loss_o.backward()
    out = net(images=x.cuda())
    torch.testing.assert_close(out["sdf_maps"].cpu().double(), out_o["sdf_maps"].detach(), atol=1e-4, rtol=0)
    loss = objectness_loss(out, gc.cuda(), gs.cuda(), sal.cuda())
    loss.backward()
    assert abs(loss.item() - loss_o.item()) < 1e-4
    nograd = net.nograd_names()
    # against the oracle's OWN ReLU decisions: whole-gradient relative L2 (a decision within rounding of zero that falls the other
    # way moves single entries by ~1e-3 * max|g|; which one does depends on the summation order -- see the test above)
    num = den = worst = 0.0
    for n, p in net.named_parameters():
        if n in nograd:
            continue
        ref = sdo[n].grad
        d = p.grad.cpu().double() - ref
        num += d.pow(2).sum().item()
        den += ref.pow(2).sum().item()
        worst = max(worst, d.abs().max().item() / (ref.abs().max().item() + 1e-12))
    assert (num / den) ** 0.5 <= 5e-4, (num / den) ** 0.5
    print(f"unmasked oracle: relative L2 {(num / den) ** 0.5:.2e}, worst entry {worst:.2e} * max|g|")
    # and on the HIP path's own linear piece (float64 oracle with its ReLU decisions imposed): rounding only, every entry
    from grad_common import masked_gradient_check
    w_inf, w_n, w_l2, n_flip = masked_gradient_check(net, sd, "dpt_tiny", x, gc, gs, sal, use_bg_sdf=bg, sdf_activation=act)
    print(f"use_bg_sdf={bg} sdf_activation={act}: worst max-norm {w_inf:.2e} ({w_n}), worst relative L2 {w_l2:.2e}, {n_flip} decisions differ")
