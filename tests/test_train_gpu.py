"""Native TrainStep (fwd + fused loss + bwd + Adam, flat parameter buffer) against the oracle's
autograd + Adam over several steps, and the extension configs' forward against the oracle."""
from argparse import Namespace

import numpy as np
import pytest
import torch

from oracle import objectness_oracle as orc
from unmore_amd import synth
from unmore_amd.hashrng import hash_init, uniform01

pytestmark = pytest.mark.gpu
ARGS = Namespace(use_bg_sdf=True, sdf_activation="tanh")


def _net(backbone, tag, dtype=torch.float32, args=ARGS):
    from unmore_amd.objectness_net import ObjectnessNet
    net = ObjectnessNet("cuda:0", 64, backbone, args)
    sd = {k: torch.from_numpy(hash_init(k, tuple(v.shape), tag)) for k, v in net.state_dict().items()}
    net.load_state_dict(sd, strict=True)
    net = net.to("cuda:0")
    net.set_compute_dtype(dtype)
    return net, sd


def test_trainstep_three_steps_match_oracle_fp32():
    from unmore_amd.trainer import TrainStep
    B, H, W = 2, 64, 96
    net, sd = _net("dpt_tiny", "tiny")
    img, cf, sdf, sal = (torch.from_numpy(a) for a in synth.make_batch(B, H, W, seed=3))
    step = TrainStep(net, lr=1e-3, lr_milestones=(2,), lr_gamma=0.1)
    sdo = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    m = {k: torch.zeros_like(v) for k, v in sd.items()}
    v_ = {k: torch.zeros_like(v) for k, v in sd.items()}
    lrs = [1e-3, 1e-3, 1e-4]  # MultiStepLR(milestones=[2], gamma=0.1), stepped per iteration
    for it in range(3):
        out5 = step.step(img.cuda(), cf.cuda(), sdf.cuda(), sal.cuda())
        for t in sdo.values():
            t.grad = None
        loss_o, terms = orc.loss_terms(orc.forward(sdo, img, orc.CONFIGS["dpt_tiny"]), cf, sdf, sal)
        loss_o.backward()
        with torch.no_grad():
            for k, t in sdo.items():
                if t.grad is not None:
                    orc.adam_update(t, t.grad, m[k], v_[k], it + 1, lr=lrs[it])
        # losses track each other step after step (the trajectories only diverge through fp32 noise)
        assert abs(out5[0].item() - loss_o.item()) < 2e-3 * max(1.0, abs(loss_o.item())), (it, out5[0].item(), loss_o.item())
        for i, t in enumerate(terms):
            assert abs(out5[1 + i].item() - t.item()) < 2e-3 * max(1.0, abs(t.item()))
    assert step.iter == 3
    # state_dict still has the reference schema and holds the updated weights
    new_sd = net.state_dict()
    assert list(new_sd.keys()) == list(sd.keys())
    moved = sum(int((new_sd[k].cpu() != sd[k]).any()) for k in sd)
    assert moved >= len(sd) - len(net.nograd_names())
    for k in net.nograd_names():
        assert torch.equal(new_sd[k].cpu(), sd[k])  # never updated, as in the reference


def test_autograd_path_with_torch_optimizer_matches_trainstep():
    """The drop-in path (loss.backward() + torch.optim.Adam, as train_objectness_net.py does) and the
    native TrainStep produce the same weights after one step."""
    from unmore_amd.loss import objectness_loss
    from unmore_amd.trainer import TrainStep
    B, H, W = 2, 64, 64
    img, cf, sdf, sal = (torch.from_numpy(a).cuda() for a in synth.make_batch(B, H, W, seed=4))
    net_a, _ = _net("dpt_tiny", "tiny")
    net_b, _ = _net("dpt_tiny", "tiny")
    opt = torch.optim.Adam(net_a.parameters(), 1e-4)
    opt.zero_grad()
    objectness_loss(net_a(images=img), cf, sdf, sal).backward()
    opt.step()
    TrainStep(net_b, lr=1e-4).step(img, cf, sdf, sal)
    for (n, pa), (_, pb) in zip(net_a.named_parameters(), net_b.named_parameters()):
        torch.testing.assert_close(pa, pb, atol=2e-6, rtol=1e-5, msg=n)
    # and the second forward of net_a sees the updated weights (packed-weight cache invalidation)
    with torch.no_grad():
        o1 = net_a(images=img)["sdf_maps"]
        o2 = net_b(images=img)["sdf_maps"]
    torch.testing.assert_close(o1, o2, atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize("backbone,H,W", [("dpt_small", 64, 96), ("dpt_large14", 70, 98)])
def test_extension_configs_match_oracle_fp32(backbone, H, W):
    """BASELINE.json extension configs: ViT-S/16 and the patch-14 wiring with odd grids
    (fusion blocks upsample to the skip's size, final upsample to the input size)."""
    net, sd = _net(backbone, backbone)
    net.eval()
    x = torch.from_numpy(uniform01(f"img:{backbone}", (1, 3, H, W)))
    with torch.no_grad():
        out = net(images=x.cuda())
        ref = orc.forward(sd, x, orc.CONFIGS[backbone])
    for k in ("center_fields", "sdf_maps"):
        assert out[k].shape == ref[k].shape
        np.testing.assert_allclose(out[k].cpu().numpy(), ref[k].numpy(), atol=1e-4, rtol=0)


def test_inference_crops_128_fp32_peaks_non_vacuous():
    """cfg5-style call: [<=50,3,128,128] crops under no_grad (object_reasoning.py:324-326) on structured images with the
    peak fixtures' weight edits (tests/peaks_common.py: plain hash weights on noise give all-zero score maps, i.e. a test that
    cannot fail).  The oracle's peak chain on the HIP fields equals the chain on the oracle's own fields wherever the score
    margin exceeds the propagated field error; every compared map must HAVE a peak."""
    import peaks_common as pc
    from unmore_amd.objectness_net import ObjectnessNet
    sd = pc.edited_state_dict(orc.state_dict_spec(orc.CONFIGS["dpt_tiny"]), "tiny", 0.05, 2.0)
    net = ObjectnessNet("cuda:0", 128, "dpt_tiny", ARGS)
    net.load_state_dict(sd, strict=True)
    net = net.to("cuda:0").eval()
    B = 6
    x = torch.from_numpy(synth.blob_images(B, 128, 128, seed=21))
    with torch.no_grad():
        out = net.get_prediction(x.cuda())
        ref = orc.forward(sd, x, orc.CONFIGS["dpt_tiny"])
    err = max((out[k].cpu() - ref[k]).abs().max().item() for k in ("sdf_maps", "center_fields"))
    assert err < 1e-4
    s_g, m_g, a_g = orc.peak_pick(out["sdf_maps"][:, 0].cpu(), out["center_fields"].cpu())
    s_r, m_r, a_r = orc.peak_pick(ref["sdf_maps"][:, 0], ref["center_fields"])
    assert (m_r > 0).sum() >= B // 2, "vacuous: no score map has a peak"
    compared = 0
    for b in range(B):
        top2 = s_r[b].flatten().topk(2).values
        if m_r[b] > 0 and (top2[0] - top2[1]).item() > 4 * err and torch.equal(s_g[b] != 0, s_r[b] != 0):
            assert a_g[b] == a_r[b], f"image {b}: argmax {int(a_g[b])} vs {int(a_r[b])}"
            compared += 1
    assert compared >= 2
    torch.testing.assert_close(m_g, m_r, atol=2 * err, rtol=0)


def test_batch_filter_matches_reference_semantics():
    """train_objectness_net.py:190-207: images whose saliency is all background or all foreground are dropped."""
    from unmore_amd.trainer import filter_batch
    B, H, W = 5, 32, 32
    img = torch.rand(B, 3, H, W).cuda()
    cf = torch.rand(B, 2, H, W).cuda()
    sdf = torch.rand(B, 1, H, W).cuda()
    sal = torch.zeros(B, 1, H, W).cuda()
    sal[1, :, 4:9, 4:9] = 1          # mixed -> kept
    sal[2] = 1                       # all foreground -> dropped
    sal[4, :, :, :16] = 1            # mixed -> kept
    i2, c2, s2, m2 = filter_batch(img, cf, sdf, sal)
    assert i2.shape[0] == 2 and torch.equal(i2, img[[1, 4]]) and torch.equal(m2, sal[[1, 4]])
    assert torch.equal(c2, cf[[1, 4]]) and torch.equal(s2, sdf[[1, 4]])


@pytest.mark.parametrize("B,H,W", [(1, 32, 32), (3, 32, 64), (1, 160, 96)])
def test_small_and_ragged_shapes_fp32(B, H, W):
    """smallest grid the reference wiring allows (2x2 patches: the stride-2 conv sees a 2x2 map), B=1, H != W."""
    net, sd = _net("dpt_tiny", "tiny")
    net.eval()
    x = torch.from_numpy(uniform01(f"img:ragged{H}x{W}", (B, 3, H, W)))
    with torch.no_grad():
        out = net(images=x.cuda())
        ref = orc.forward(sd, x, orc.CONFIGS["dpt_tiny"])
    for k in ("center_fields", "sdf_maps"):
        np.testing.assert_allclose(out[k].cpu().numpy(), ref[k].numpy(), atol=1e-4, rtol=0)


@pytest.mark.parametrize("dtype,H,W", [(torch.float32, 64, 96), (torch.bfloat16, 64, 64)])
def test_collapsed_sdf_head_equals_factored(dtype, H, W):
    """Opt-in algebraic form of the linear boundary-distance head: same outputs and the same gradients for EVERY
    factored weight (fp32: vs the fp64 oracle at the bar of the factored path; bf16: vs the factored bf16 path)."""
    from unmore_amd.loss import objectness_loss
    B = 2
    img, cf, sdf, sal = (torch.from_numpy(a) for a in synth.make_batch(B, H, W, seed=5))
    nets = {}
    for mode in ("factored", "collapsed"):
        net, sd = _net("dpt_tiny", "tiny", dtype)
        net.set_sdf_head_mode(mode)
        net.train()
        out = net(images=img.cuda())
        loss = objectness_loss(out, cf.cuda(), sdf.cuda(), sal.cuda())
        loss.backward()
        nets[mode] = (net, out, loss.item())
    o_f, o_c = nets["factored"][1], nets["collapsed"][1]
    tol = 1e-4 if dtype == torch.float32 else 3e-2
    assert (o_f["sdf_maps"] - o_c["sdf_maps"]).abs().max().item() < tol
    assert torch.equal(o_f["center_fields"], o_c["center_fields"])
    if dtype == torch.float32:
        # both paths against each other, parameter by parameter (a comparison with the fp64 oracle would also measure
        # ReLU masks that fp32 and fp64 decide differently -- identical for the two paths, unrelated to the algebra)
        assert abs(nets["collapsed"][2] - nets["factored"][2]) < 1e-5
        gf = dict((n, p.grad) for n, p in nets["factored"][0].named_parameters())
        for n, p in nets["collapsed"][0].named_parameters():
            if gf[n] is None:
                assert p.grad is None
                continue
            err = (p.grad - gf[n]).abs().max().item() / (gf[n].abs().max().item() + 1e-12)
            assert err < 2e-4, (n, err)
        # and the head's own gradients against the fp64 oracle
        sdo = {k: v.clone().double().requires_grad_(True) for k, v in sd.items()}
        lo, _ = orc.loss_terms(orc.forward(sdo, img.double(), orc.CONFIGS["dpt_tiny"]), cf.double(), sdf.double(), sal.double())
        lo.backward()
        for n, p in nets["collapsed"][0].named_parameters():
            if n.startswith("sdf_prediction_head"):
                ref = sdo[n].grad
                err = (p.grad.cpu().double() - ref).abs().max().item() / (ref.abs().max().item() + 1e-12)
                assert err < 5e-4, (n, err)
    else:
        a = torch.cat([p.grad.flatten() for _, p in nets["factored"][0].named_parameters() if p.grad is not None])
        b = torch.cat([p.grad.flatten() for _, p in nets["collapsed"][0].named_parameters() if p.grad is not None])
        assert torch.dot(a, b) / (a.norm() * b.norm()) > 0.99


@pytest.mark.parametrize("dtype,H,W", [(torch.float32, 64, 96), (torch.bfloat16, 64, 64), (torch.float32, 64, 32), (torch.float32, 96, 160)])
def test_linear_head_algebraic_backward_equals_gemm_backward(dtype, H, W):
    """The boundary-distance head's two BACKWARD forms behind the same four-convolution forward ('factored' mode; since round 6 the
    default mode collapses that forward when the backward is algebraic, tests/test_collapsed_train_gpu.py): its backward
    takes the exact gradients of all eight factored tensors from three pixel reductions instead of layer-by-layer GEMMs
    (engine._linear_head_backward).  Both backward forms against each other for EVERY parameter of the net (fp32: 2e-4 *
    max|g|; the fp64-oracle bar of 5e-4 is asserted on the default path by tests/test_model_gpu.py); bf16: cosine."""
    from unmore_amd.loss import objectness_loss
    B = 2
    img, cf, sdf, sal = (torch.from_numpy(a) for a in synth.make_batch(B, H, W, seed=6))
    res = {}
    for mode in ("gemm", "algebraic"):
        net, _ = _net("dpt_tiny", "tiny", dtype)
        net.set_sdf_head_mode("factored")
        net.set_linear_head_backward(mode)
        net.train()
        out = net(images=img.cuda())
        objectness_loss(out, cf.cuda(), sdf.cuda(), sal.cuda()).backward()
        res[mode] = (out, {n: p.grad for n, p in net.named_parameters()})
    if dtype == torch.float32:
        # fp32 mode: a head without saved activations runs its forward on the bf16-plane kernel (same six-term fp32-grade products,
        # different tiling: equal to rounding, not bit for bit)
        torch.testing.assert_close(res["gemm"][0]["sdf_maps"], res["algebraic"][0]["sdf_maps"], atol=3e-6, rtol=0)
    else:
        assert torch.equal(res["gemm"][0]["sdf_maps"], res["algebraic"][0]["sdf_maps"])       # same forward, bit for bit
    assert torch.equal(res["gemm"][0]["center_fields"], res["algebraic"][0]["center_fields"])
    gg, ga = res["gemm"][1], res["algebraic"][1]
    if dtype == torch.float32:
        for n in gg:
            if gg[n] is None:
                assert ga[n] is None
                continue
            err = (ga[n] - gg[n]).abs().max().item() / (gg[n].abs().max().item() + 1e-12)
            assert err < 2e-4, (n, err)
    else:
        a = torch.cat([g.flatten() for g in gg.values() if g is not None]).double()
        b = torch.cat([ga[n].flatten() for n, g in gg.items() if g is not None]).double()
        assert torch.dot(a, b) / (a.norm() * b.norm()) > 0.995


def test_fused_head_output_layer_matches_unfused(monkeypatch):
    """At >= 2048 output tiles the heads' last layer rides in the epilogue of the GEMM before it (engine.py heads section,
    umr_gemm_desc.red_*): same outputs and gradients as the stand-alone head_out_fwd kernel, in training (h3 saved) and
    inference (h3 never stored) form."""
    from unmore_amd import engine, ops
    net, _ = _net("dpt_tiny", "tiny", torch.bfloat16)
    net.set_sdf_head_mode("factored")     # the four-convolution form of BOTH heads at inference too (the default collapses the sdf head there)
    x = torch.from_numpy(uniform01("img:fuse", (2, 3, 256, 256))).cuda()
    h2 = torch.zeros((2 * 256 * 256, 512), dtype=torch.bfloat16, device="cuda:0")
    w3 = torch.zeros((1024, 512), dtype=torch.bfloat16, device="cuda:0")
    assert ops.gemm_nt(h2, w3, torch.zeros(1024, device="cuda:0"), query_rowreduce=True), "size chosen to take the fused path"

    def run(fuse):
        monkeypatch.setattr(engine, "_FUSE_HEAD_OUT", fuse)
        net.zero_grad(set_to_none=True)
        out = net(images=x)
        (out["center_fields"].square().mean() + out["sdf_maps"].abs().mean()).backward()
        grads = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
        with torch.no_grad():
            inf = net.get_prediction(x)
        return out, grads, inf

    o1, g1, i1 = run(True)
    o0, g0, i0 = run(False)
    for k in ("center_fields", "sdf_maps"):
        torch.testing.assert_close(o1[k], o0[k], atol=2e-3, rtol=0)     # bf16 h3 is identical; only the f32 summation order differs
        torch.testing.assert_close(i1[k], o1[k].detach(), atol=0, rtol=0)  # no_store form = training form
        torch.testing.assert_close(i0[k], o0[k].detach(), atol=0, rtol=0)
    assert g1.keys() == g0.keys()
    for n in g1:
        scale = float(g0[n].abs().max()) + 1e-12
        assert float((g1[n] - g0[n]).abs().max()) <= 2e-2 * scale, n


def test_bf16_training_tracks_fp32_and_descends():
    """throughput mode (bf16 storage, fp32 accumulation, fp32 master weights) against parity mode over 25 steps of the
    native train step on the same data: both losses fall, and stay within a few percent of each other"""
    from unmore_amd.trainer import TrainStep
    B, H, W = 4, 64, 64
    img, cf, sdf, sal = (torch.from_numpy(a).cuda() for a in synth.make_batch(B, H, W, seed=5))
    curves = {}
    for dt in (torch.float32, torch.bfloat16):
        net, _ = _net("dpt_tiny", "tiny", dt)
        step = TrainStep(net, lr=2e-5)   # small steps: Adam's sign-like updates make larger ones chaotic on this toy problem
        curves[dt] = [float(step.step(img, cf, sdf, sal)[0]) for _ in range(25)]
    f32, b16 = curves[torch.float32], curves[torch.bfloat16]
    assert all(np.isfinite(f32)) and all(np.isfinite(b16))
    assert f32[-1] < f32[0] and b16[-1] < b16[0], (f32[0], f32[-1], b16[0], b16[-1])
    for a, b in zip(f32, b16):
        assert abs(a - b) <= 0.02 * abs(a) + 1e-3, (a, b)


def test_large_shape_determinism_and_batch_independence():
    """size-independent properties at a shape that runs on the 256x256 kernels (no CPU oracle at this size): two runs give
    bitwise identical outputs and gradients (every reduction is fixed-order, no float atomics), and an image's maps do
    not depend on what else is in the batch"""
    net, _ = _net("dpt_tiny", "tiny", torch.bfloat16)
    x = torch.from_numpy(uniform01("img:props", (3, 3, 256, 256))).cuda()

    def run(inp):
        net.zero_grad(set_to_none=True)
        out = net(images=inp)
        (out["center_fields"].square().mean() + out["sdf_maps"].abs().mean()).backward()
        return out, {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}

    o1, g1 = run(x)
    o2, g2 = run(x)
    for k in ("center_fields", "sdf_maps"):
        assert torch.equal(o1[k], o2[k]), k
    for n in g1:
        assert torch.equal(g1[n], g2[n]), n
    with torch.no_grad():
        single = net.get_prediction(x[1:2])
    for k in ("center_fields", "sdf_maps"):
        torch.testing.assert_close(single[k], o1[k][1:2].detach(), atol=3e-2, rtol=0)   # other kernels / summation orders at B=1


def test_checkpoint_resume_matches_uninterrupted_run_and_torch_adam_format():
    """train_objectness_net.py:118-123,268-275: model + optimizer state + iteration saved after 2 steps and loaded into a fresh
    model / TrainStep give bit-identical weights after step 3; the optimizer state has torch.optim.Adam's schema."""
    from argparse import Namespace
    from unmore_amd import synth
    from unmore_amd.hashrng import hash_init
    from unmore_amd.objectness_net import ObjectnessNet
    from unmore_amd.trainer import TrainStep

    def make():
        net = ObjectnessNet("cuda:0", 64, "dpt_tiny", Namespace(use_bg_sdf=True, sdf_activation="tanh"))
        net.load_state_dict({k: torch.from_numpy(hash_init(k, tuple(v.shape), "tiny")) for k, v in net.state_dict().items()})
        return net.to("cuda:0")

    batches = [tuple(torch.from_numpy(a).cuda() for a in synth.make_batch(2, 64, 64, seed=s)) for s in (1, 2, 3)]
    a = make()
    sa = TrainStep(a, lr=1e-3, lr_milestones=(2,), lr_gamma=0.5)
    for b in batches[:2]:
        sa.step(*b)
    ckpt = {"model_state_dict": {k: v.detach().cpu().clone() for k, v in a.state_dict().items()},
            "optimizer_state_dict": sa.optimizer_state_dict(), "iter": sa.iter}
    sa.step(*batches[2])
    # torch's own optimizer accepts the state (schema check): same parameter order, same per-parameter keys
    ref_params = [torch.nn.Parameter(p.detach().cpu().clone()) for p in a.parameters()]
    opt = torch.optim.Adam(ref_params, lr=1e-3)
    opt.load_state_dict({"state": {i: {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in st.items()} for i, st in ckpt["optimizer_state_dict"]["state"].items()},
                         "param_groups": ckpt["optimizer_state_dict"]["param_groups"]})
    assert len(opt.state_dict()["state"]) == len(ckpt["optimizer_state_dict"]["state"]) > 0
    b_ = make()
    b_.load_state_dict(ckpt["model_state_dict"], strict=True)
    sb = TrainStep(b_, lr=1e-3, lr_milestones=(2,), lr_gamma=0.5)
    sb.load_optimizer_state_dict(ckpt["optimizer_state_dict"], iteration=ckpt["iter"])
    assert sb.iter == 2 and sb.current_lr_for_step() == sa.lr0  # (about to run step 3: lr of 2 completed steps)
    sb.step(*batches[2])
    for (n, p), (_, q) in zip(a.named_parameters(), b_.named_parameters()):
        assert torch.equal(p, q), n


def test_empty_batch_is_handled():
    """the batch filter can leave nothing (train_objectness_net.py:190-207): forward returns empty maps like the reference's
    convolutions do; the native step refuses instead of feeding NaN means to Adam"""
    from unmore_amd.trainer import TrainStep
    net, _ = _net("dpt_tiny", "tiny")
    x = torch.zeros((0, 3, 64, 64), device="cuda:0")
    with torch.no_grad():
        out = net.get_prediction(x)
    assert out["center_fields"].shape == (0, 2, 64, 64) and out["sdf_maps"].shape == (0, 1, 64, 64)
    step = TrainStep(net)
    with pytest.raises(ValueError, match="empty batch"):
        step.step(x, torch.zeros((0, 2, 64, 64), device="cuda:0"), torch.zeros((0, 1, 64, 64), device="cuda:0"), torch.zeros((0, 1, 64, 64), device="cuda:0"))


def test_sizes_that_are_not_multiples_of_the_patch_follow_the_reference():
    """100x100 with patch 16: the token grid is 6x6 (the remainder is ignored by the strided patch conv, vit.py:179) and every
    upsampling is exactly x2 (blocks.py:377-379, models.py:70-72), so the maps are 96x96 -- as the reference's.  An odd grid
    (48x48 -> 3x3) cannot pass the skip addition (blocks.py:372) and raises there too."""
    net, sd = _net("dpt_tiny", "tiny")
    net.eval()
    x = torch.from_numpy(uniform01("img:odd100", (1, 3, 100, 100)))
    with torch.no_grad():
        out = net(images=x.cuda())
        ref = orc.forward(sd, x, orc.CONFIGS["dpt_tiny"])
    assert ref["sdf_maps"].shape == (1, 1, 96, 96) and out["sdf_maps"].shape == (1, 1, 96, 96)
    for k in ("center_fields", "sdf_maps"):
        np.testing.assert_allclose(out[k].cpu().numpy(), ref[k].numpy(), atol=1e-4, rtol=0)
    with pytest.raises((AssertionError, RuntimeError)):
        with torch.no_grad():
            net(images=torch.zeros(1, 3, 48, 48, device="cuda:0"))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_batched_repack_equals_lazy_repack(dtype, monkeypatch):
    """After the Adam launch every kernel-layout weight copy is refreshed by ONE umr_permute4_batched launch (PackCache.refresh)
    instead of being dropped and re-packed weight by weight: four steps either way end in bit-identical weights and losses."""
    from unmore_amd import trainer
    B, H, W = 2, 64, 64
    img, cf, sdf, sal = (torch.from_numpy(a).cuda() for a in synth.make_batch(B, H, W, seed=4))
    res = {}
    for mode in (True, False):
        monkeypatch.setattr(trainer, "_BATCHED_REPACK", mode)
        net, _ = _net("dpt_tiny", "tiny", dtype)
        step = trainer.TrainStep(net, lr=1e-3)
        # (eight steps: from the third on the step is replayed as a chain of graphs on two lanes -- with the lazy re-pack the whole Adam
        # update runs on the main lane AFTER the join of the weight-gradient lane, a join the chain lacked at first: this test caught it)
        losses = [step.step(img, cf, sdf, sal).clone() for _ in range(8)]
        if mode:
            eng = net._engine()
            # (one launch for all copies, or -- small problems -- one per stage behind the weight-gradient stream)
            assert eng.cache._replay and sum(len(v[-1]) for v in eng.cache._replay.values() if v) > 40 and not eng.cache._o
        res[mode] = (torch.stack(losses).cpu(), step.flat_p.clone().cpu())
    assert torch.equal(res[True][0], res[False][0])
    assert torch.equal(res[True][1], res[False][1])


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("graph", ["off", "auto"])
def test_adam_pack_equals_adam_then_refresh(dtype, graph, monkeypatch):
    """Round 6: per stage, ONE optimizer launch that writes the bf16 [N,K] / [K,N] copies of the stage's Linear weights itself
    (trainer._ADAM_PACK, PackCache.adam_and_refresh, umr_adam_pack_step) against the Adam launch followed by the batched refresh: ten
    steps end in bit-identical losses, weights, Adam moments AND packed copies -- eagerly on two streams and replayed as a chain of
    graphs.  (fp32 mode has no bf16 copies: the fused form declines and both arms run the same launches.)"""
    from unmore_amd import trainer
    B, H, W = 2, 64, 64
    res = {}
    for fused in (True, False):
        monkeypatch.setattr(trainer, "_ADAM_PACK", fused)
        net, _ = _net("dpt_tiny", "tiny", dtype)
        step = trainer.TrainStep(net, lr=1e-3).set_graph_mode(graph)
        losses = []
        for it in range(10):
            img, cf, sdf, sal = (torch.from_numpy(a).cuda() for a in synth.make_batch(B, H, W, seed=60 + it))
            losses.append(step.step(img, cf, sdf, sal).clone())
        torch.cuda.synchronize()
        eng = net._engine()
        packs = {k: e[1].clone() for k, e in eng.cache._c.items() if torch.is_tensor(e[1])}
        n_fused = sum(1 for k, v in eng.cache._replay.items() if isinstance(k, tuple) and k[0] == "adam" and v)
        assert (n_fused > 0) == (fused and dtype == torch.bfloat16), n_fused
        res[fused] = (torch.stack(losses).cpu(), step.flat_p.clone(), step.m.clone(), step.v.clone(), packs)
    a, b = res[True], res[False]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
    assert a[4].keys() == b[4].keys() and len(a[4]) > 20
    for k in a[4]:
        assert torch.equal(a[4][k], b[4][k]), k
    # and the copies ARE the weights: every bf16 [N,K] copy equals the cast of its parameter after the last update
    P = dict(net.named_parameters())
    n_chk = 0
    for key, t in a[4].items():
        if len(key) != 3 or key[0] not in P:      # (fp32 mode keys its plane packs differently)
            continue
        name, kind, dt_ = key
        if kind == "lin" and dt_ == torch.bfloat16 and P[name].dim() == 2:
            assert torch.equal(t, P[name].detach().to(torch.bfloat16)), name
            n_chk += 1
        if kind == "lin_t" and dt_ == torch.bfloat16 and P[name].dim() == 2:
            assert torch.equal(t, P[name].detach().t().to(torch.bfloat16)), name
            n_chk += 1
    assert n_chk > 10 or dtype == torch.float32


def test_adam_pack_kernel_against_adam_and_casts():
    """umr_adam_pack_step alone: plain ranges (with a tail that is not a multiple of four) and weight entries with one, the other or both
    copies, ragged 64 x 64 tiles -- against umr_adam_step_hyper on the same buffers and torch casts / transposes of the result"""
    from unmore_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(1)
    shapes = [("plain", 1003), ("weight", (72, 100), True, True), ("plain", 4096 + 8), ("weight", (192, 64), True, False),
              ("weight", (64, 256), False, True), ("plain", 3), ("weight", (136, 68), True, True)]
    total = sum((sh[1] if sh[0] == "plain" else sh[1][0] * sh[1][1] + 0) for sh in shapes)
    total = (total + 63) // 64 * 64 + 64 * len(shapes)
    bufs = [torch.randn(total, generator=g).to(dev) for _ in range(3)] + [torch.rand(total, generator=g).to(dev)]
    ref = [b.clone() for b in bufs]
    hyper = torch.zeros(8, device=dev)
    ops.adam_set_hyper(hyper, 3, 1e-3, 0.9, 0.999, 1e-8, 0.5)
    entries, copies, off = [], [], 0
    for sh in shapes:
        if sh[0] == "plain":
            n = sh[1]
            entries.append(("plain",) + tuple(b[off:off + n] for b in bufs))
            off += (n + 63) // 64 * 64
        else:
            N, K = sh[1]
            dl = torch.zeros((N, K), dtype=torch.bfloat16, device=dev) if sh[2] else None
            dt_ = torch.zeros((K, N), dtype=torch.bfloat16, device=dev) if sh[3] else None
            entries.append(("weight",) + tuple(b[off:off + N * K].view(N, K) for b in bufs) + (dl, dt_))
            copies.append((off, N, K, dl, dt_))
            off += N * K
    ops.adam_pack(entries, hyper)()
    # reference: the plain Adam launch over exactly the ranges the entries cover
    for ent in entries:
        sl = [t.reshape(-1) for t in ent[1:5]]
        o = sl[0].data_ptr() - bufs[0].data_ptr()
        o //= 4
        n = sl[0].numel()
        ops.adam_step_hyper(ref[0][o:o + n], ref[1][o:o + n], ref[2][o:o + n], ref[3][o:o + n], hyper)
    torch.cuda.synchronize()
    for a, b in zip(bufs, ref):
        assert torch.equal(a, b)
    for off, N, K, dl, dt_ in copies:
        w = bufs[0][off:off + N * K].view(N, K)
        if dl is not None:
            assert torch.equal(dl, w.to(torch.bfloat16))
        if dt_ is not None:
            assert torch.equal(dt_, w.t().to(torch.bfloat16))


def test_permute4_batched_equals_single_launches():
    """umr_permute4_batched (linear and LDS-tiled block shapes, ragged edges) against one umr_permute4 launch per tensor: the
    engine's own pack recipes -- conv weights to [co][ky][kx][ci], the flipped data-gradient form [ci][2-ky][2-kx][co], Linear
    transposes of row-strided slices, bias replication, plain casts."""
    from unmore_amd import engine, ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    recipes, refs = [], []

    def record(build):
        rec = []
        ops._pack_recorder = rec
        try:
            ref = build()
        finally:
            ops._pack_recorder = None
        assert len(rec) == 1
        src, dst, dims, strides, off = rec[0]
        refs.append(ref.clone())
        recipes.append((src, torch.zeros_like(dst), dims, strides, off))

    for co, ci in [(8, 16), (33, 7), (256, 64), (130, 70), (512, 512)]:
        w = torch.randn((co, ci, 3, 3), generator=g).to(dev)
        for dt in (torch.float32, torch.bfloat16):
            record(lambda: engine._pack_conv3(w, dt))
            record(lambda: engine._pack_conv3_dgrad(w, dt))
    wl = torch.randn((300, 1536), generator=g).to(dev)
    record(lambda: engine._pack_linear(wl, torch.bfloat16))
    record(lambda: engine._pack_linear_t(wl, torch.bfloat16))
    record(lambda: engine._pack_linear_t(wl[:, :768], torch.bfloat16))       # row-strided slice (the readout projection halves)
    record(lambda: engine._pack_linear_t(wl[:, 768:], torch.float32))
    # the 16-byte form of the 64x64 transpose (R % 4 == 0, C % 8 == 0, aligned): full tiles and ragged edges in both directions
    for n_, k_ in ((1024, 4096), (200, 1000), (72, 260), (3072, 768)):
        wv = torch.randn((n_, k_), generator=g).to(dev)
        record(lambda: engine._pack_linear_t(wv, torch.bfloat16))
    big8 = torch.randn(8192 * 5 + 2048 * 3, generator=g).to(dev)      # whole 8192-element blocks of a cast + a tail of whole 2048-runs
    record(lambda: ops.cast(big8, torch.bfloat16))
    wt = torch.randn((96, 96, 4, 4), generator=g).to(dev)
    record(lambda: engine._pack_convT(wt, torch.bfloat16))
    record(lambda: engine._pack_convT_dgrad(wt, torch.bfloat16))
    record(lambda: engine._rep_bias(torch.randn(96, generator=g).to(dev), 16))
    big = torch.randn(300001, generator=g).to(dev)      # a cast (1-D) with a ragged tail, more than one block
    record(lambda: ops.cast(big, torch.bfloat16))
    kinds = [ops._perm_tile(r[2], r[3]) for r in recipes]
    assert sum(k is not None for k in kinds) >= 10 and any(k is None for k in kinds)
    assert sum(k == "t2d" for k in kinds) >= 3 and any(isinstance(k, tuple) for k in kinds)   # 64x64 transposes and generic tiles
    launch = ops.permute4_batched(recipes)
    launch()
    torch.cuda.synchronize()
    for i, ((src, dst, *_), ref) in enumerate(zip(recipes, refs)):
        assert torch.equal(dst, ref), i
