"""Round 6: the boundary-distance head's collapsed forward as the TRAINING default.

The head (objectness_net.py:128-135; tanh / None variants, objectness_net.py:119-142) has no non-linearity between its four
convolutions, and its default backward (engine._linear_head_backward, 'algebraic') reads the head's OUTPUT only.  So in 'auto' mode
a training step evaluates the head as one 3x3 conv 256 -> 1 (a 16-column tap GEMM on the map before the final resize), exactly as
inference calls have since round 5; the weights stay factored (same state_dict schema, same eight gradients).  Qualification:

  * mode decisions: tanh / None collapse in training iff the backward is algebraic; 'sine' (its backward needs the pre-activation)
    and every ReLU variant keep the four convolutions, bit-identical to 'factored' mode;
  * three TrainStep steps in the default mode on dpt_tiny and dpt_base 128x128, tanh and None: at every step the loss is within
    1e-4 of the float64 oracle's and every parameter gradient within 5e-5 (max-norm and relative L2) on the HIP path's own linear
    piece (tests/grad_common.py -- the suite's bars, unchanged);
  * bf16: the first-step gradient of the collapsed form is as close to the fp32 factored gradient as the factored bf16 form is."""
from argparse import Namespace

import pytest
import torch

from oracle import objectness_oracle as orc
from unmore_amd import synth
from unmore_amd.hashrng import hash_init, uniform01

pytestmark = pytest.mark.gpu


def _net(backbone, tag, args, dtype=torch.float32, size=64, mode=None, bwd=None):
    from unmore_amd.objectness_net import ObjectnessNet
    net = ObjectnessNet("cuda:0", size, backbone, args)
    sd = {k: torch.from_numpy(hash_init(k, tuple(v.shape), tag)) for k, v in net.state_dict().items()}
    net.load_state_dict(sd, strict=True)
    net = net.to("cuda:0")
    net.set_compute_dtype(dtype)
    if mode is not None:
        net.set_sdf_head_mode(mode)
    if bwd is not None:
        net.set_linear_head_backward(bwd)
    return net, sd


def test_auto_mode_decisions_in_training():
    A = lambda act, bg=True: Namespace(use_bg_sdf=bg, sdf_activation=act)
    for act in ("tanh", None):
        eng = _net("dpt_tiny", "tiny", A(act))[0]._engine()
        assert eng.collapse_linear_heads == "auto"
        assert eng._collapse(eng.sdf_layout, save=True) and eng._collapse(eng.sdf_layout, save=False)
        assert not eng._collapse(eng.center_layout, save=True) and not eng._collapse(eng.center_layout, save=False)
        # a layer-by-layer GEMM backward reads the four convolutions' activations: the forward keeps them
        eng = _net("dpt_tiny", "tiny", A(act), bwd="gemm")[0]._engine()
        assert not eng._collapse(eng.sdf_layout, save=True) and eng._collapse(eng.sdf_layout, save=False)
        eng = _net("dpt_tiny", "tiny", A(act), mode="factored")[0]._engine()
        assert not eng._collapse(eng.sdf_layout, save=True) and not eng._collapse(eng.sdf_layout, save=False)
    eng = _net("dpt_tiny", "tiny", A("sine"))[0]._engine()
    assert not eng._collapse(eng.sdf_layout, save=True) and eng._collapse(eng.sdf_layout, save=False)
    for args in (A("relu"), A("tanh", bg=False), A(None, bg=False)):
        eng = _net("dpt_tiny", "tiny", args)[0]._engine()
        assert not eng._collapse(eng.sdf_layout, save=True) and not eng._collapse(eng.sdf_layout, save=False)


@pytest.mark.parametrize("act,bg,bwd", [("sine", True, None), ("relu", True, None), ("tanh", False, None), ("tanh", True, "gemm")])
def test_variants_that_must_stay_factored_train_bit_identically_to_factored_mode(act, bg, bwd):
    """outputs AND every parameter gradient of a training step in the default mode equal those of 'factored' mode bit for bit"""
    from unmore_amd.loss import objectness_loss
    args = Namespace(use_bg_sdf=bg, sdf_activation=act)
    img, cf, sdf, sal = (torch.from_numpy(a).cuda() for a in synth.make_batch(2, 64, 96, seed=21))
    res = {}
    for mode in ("auto", "factored"):
        net, _ = _net("dpt_tiny", "tiny", args, mode=mode, bwd=bwd)
        net.train()
        out = net(images=img)
        objectness_loss(out, cf, sdf, sal).backward()
        res[mode] = (out, {n: p.grad for n, p in net.named_parameters()})
    for k in ("center_fields", "sdf_maps"):
        assert torch.equal(res["auto"][0][k], res["factored"][0][k]), k
    for n, g in res["factored"][1].items():
        ga = res["auto"][1][n]
        assert (g is None and ga is None) or torch.equal(g, ga), n


@pytest.mark.parametrize("backbone,tag,B,H,W,act", [("dpt_tiny", "tiny", 2, 64, 96, "tanh"), ("dpt_tiny", "tiny", 2, 64, 96, None),
                                                     ("dpt_base", "base", 2, 128, 128, "tanh"), ("dpt_base", "base", 1, 128, 96, None)])
def test_three_default_train_steps_against_the_float64_oracle(backbone, tag, B, H, W, act):
    """TrainStep in the default mode (collapsed forward, algebraic backward).  Before each of three optimizer steps the net's CURRENT
    weights go through tests/grad_common.masked_gradient_check: loss within 1e-4 of the float64 oracle's, both maps within 1e-4,
    every parameter gradient within 5e-5 * max|g| and 5e-5 relative L2 on the HIP path's linear piece.  Then the step's own loss is
    held to the same oracle loss, and the weights must have moved."""
    from grad_common import masked_gradient_check
    from unmore_amd.trainer import TrainStep
    args = Namespace(use_bg_sdf=True, sdf_activation=act)
    net, sd0 = _net(backbone, tag, args, size=H)
    net.train()
    eng = net._engine()
    assert eng._collapse(eng.sdf_layout, save=True)
    _, cf, sdf, sal = (torch.from_numpy(a) for a in synth.make_batch(B, H, W, seed=13))
    img = torch.from_numpy(synth.blob_images(B, H, W, seed=13))
    # the reference's learning rate (train_objectness_net.py:96): Adam's first steps move EVERY weight by ~lr whatever its gradient, and
    # at 1e-3 this random-init net's centre field runs off to |values| ~ 60 within one step, where the absolute 1e-4 bars mean 2e-6 relative
    step = TrainStep(net, lr=1e-4, lr_milestones=(2,), lr_gamma=0.1).set_graph_mode("off")
    prev = None
    for it in range(3):
        sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
        if prev is not None:
            assert any(not torch.equal(sd[k], prev[k]) for k in sd)
        if backbone == "dpt_tiny" or it != 1:       # (dpt_base: the float64 oracle's forward + backward take ~20 s; first and last step)
            w_inf, w_n, w_l2, n_flip = masked_gradient_check(net, sd, backbone, img, cf, sdf, sal, use_bg_sdf=True, sdf_activation=act)
            print(f"{backbone} {act} step {it}: worst max-norm {w_inf:.2e} ({w_n}), worst relative L2 {w_l2:.2e}, {n_flip} ReLU decisions differ")
        sdo = {k: v.double() for k, v in sd.items()}
        with torch.no_grad():
            loss_o, terms = orc.loss_terms(orc.forward(sdo, img.double(), orc.CONFIGS[backbone], use_bg_sdf=True, sdf_activation=act),
                                           cf.double(), sdf.double(), sal.double())
        out5 = step.step(img.cuda(), cf.cuda(), sdf.cuda(), sal.cuda())
        assert abs(out5[0].item() - loss_o.item()) < 1e-4, (it, out5[0].item(), loss_o.item())
        for i, t in enumerate(terms):
            assert abs(out5[1 + i].item() - t.item()) < 1e-4, (it, i)
        prev = sd
    assert step.iter == 3


def test_default_train_step_equals_explicit_collapsed_mode_and_tracks_factored():
    """'auto' in training IS 'collapsed' (bit for bit); against 'factored' the outputs differ by rounding only (fp32: 2e-5) and the
    gradients of every parameter agree to 2e-4 * max|g| (the bar of test_train_gpu.py::test_collapsed_sdf_head_equals_factored)"""
    from unmore_amd.trainer import TrainStep
    args = Namespace(use_bg_sdf=True, sdf_activation="tanh")
    img, cf, sdf, sal = (torch.from_numpy(a).cuda() for a in synth.make_batch(2, 64, 96, seed=5))
    G, L = {}, {}
    for mode in ("auto", "collapsed", "factored"):
        net, _ = _net("dpt_tiny", "tiny", args, mode=mode)
        net.train()
        step = TrainStep(net, lr=0.0).set_graph_mode("off")
        L[mode] = step.step(img, cf, sdf, sal).cpu()
        G[mode] = {n: t.clone() for n, t in step.G.items()}
    assert torch.equal(L["auto"], L["collapsed"])
    for n in G["auto"]:
        assert torch.equal(G["auto"][n], G["collapsed"][n]), n
        gf = G["factored"][n]
        assert (G["auto"][n] - gf).abs().max().item() <= 2e-4 * (gf.abs().max().item() + 1e-12), n
    assert (L["auto"] - L["factored"]).abs().max().item() < 1e-5


def _first_step_grads(backbone, tag, H, W, B, dtype, mode):
    from unmore_amd.trainer import TrainStep
    args = Namespace(use_bg_sdf=True, sdf_activation="tanh")
    _, cf, sdf, sal = (torch.from_numpy(a).cuda() for a in synth.make_batch(B, H, W, seed=9))
    img = torch.from_numpy(synth.blob_images(B, H, W, seed=9)).cuda()
    net, _ = _net(backbone, tag, args, dtype=dtype, size=H, mode=mode)
    net.train()
    step = TrainStep(net, lr=0.0).set_graph_mode("off")
    out5 = step.step(img, cf, sdf, sal).cpu()
    g = torch.cat([t.flatten() for t in step.G.values()]).double().cpu()
    per = {n: t.double().flatten().cpu() for n, t in step.G.items() if n.startswith("sdf_prediction_head")}
    del step, net
    torch.cuda.empty_cache()
    return out5, g, per


def test_bf16_collapsed_first_step_gradient_is_as_close_to_fp32_as_the_factored_form():
    """dpt_base 384x384 B = 8 (the benchmark's kernels: 256x256 persistent GEMMs, fused head reduction): the bf16 gradient of the
    default (collapsed) step and of the factored step, both against the fp32 FACTORED gradient -- the collapsed form must not be
    further from it than the factored bf16 form is (cosine within 2e-5 of it or better), globally and for the head's own eight tensors."""
    cos = lambda a, b: (torch.dot(a, b) / (a.norm() * b.norm() + 1e-300)).item()
    l32, g32, p32 = _first_step_grads("dpt_base", "base", 384, 384, 8, torch.float32, "factored")
    lf, gf, pf = _first_step_grads("dpt_base", "base", 384, 384, 8, torch.bfloat16, "factored")
    lc, gc, pcol = _first_step_grads("dpt_base", "base", 384, 384, 8, torch.bfloat16, "auto")
    c_f, c_c = cos(gf, g32), cos(gc, g32)
    print(f"bf16 first-step gradient vs fp32 factored at dpt_base 384x384 B=8: factored cosine {c_f:.6f}, collapsed (default) cosine {c_c:.6f}; "
          f"loss fp32 {l32[0].item():.5f}, bf16 factored {lf[0].item():.5f}, bf16 collapsed {lc[0].item():.5f}")
    assert c_c > 0.999 and c_c >= c_f - 2e-5
    assert abs(lc[0].item() - l32[0].item()) <= max(2e-2, abs(lf[0].item() - l32[0].item()) + 2e-3)
    for n in p32:
        cf_, cc_ = cos(pf[n], p32[n]), cos(pcol[n], p32[n])
        print(f"  {n}: factored {cf_:.6f} collapsed {cc_:.6f}")
        assert cc_ >= min(cf_, 0.9999) - 1e-3, (n, cf_, cc_)
