"""Qualification of the inference default of the boundary-distance head (set_sdf_head_mode('auto'): collapsed under no_grad).

The head has no non-linearity before its tanh (reference models/objectness_net.py:128-135), so its four convolutions compose to one
3x3 conv 256 -> 1 plus a border-dependent bias (legal for inference if the 1e-4 contract holds).  Here the
collapsed form is held to exactly the bars of the factored form: every reference-made forward fixture at 1e-4 in fp32, the peak
chain (HIP fp32 net -> HIP peaks vs reference net -> reference peaks, bit-exact indices on certified maps), all head variants
against the float64 oracle, and factored-vs-collapsed directly.  Training keeps the factored form by default (checked)."""
import os
from argparse import Namespace

import numpy as np
import pytest
import torch

import peaks_common as pc
from oracle import objectness_oracle as orc
from unmore_amd import synth
from unmore_amd.hashrng import hash_init, uniform01

pytestmark = pytest.mark.gpu
ARGS = Namespace(use_bg_sdf=True, sdf_activation="tanh")


def _net(backbone, tag=None, sd=None, dtype=torch.float32, size=128, args=ARGS, mode="collapsed"):
    from unmore_amd.objectness_net import ObjectnessNet
    net = ObjectnessNet("cuda:0", size, backbone, args)
    if sd is None:
        sd = {k: torch.from_numpy(hash_init(k, tuple(v.shape), tag)) for k, v in net.state_dict().items()}
    net.load_state_dict(sd, strict=True)
    net = net.to("cuda:0").to(torch.float32)
    net.set_compute_dtype(dtype)
    net.set_sdf_head_mode(mode)
    net.eval()
    return net, sd


FULL = [
    ("dpt_tiny", "tiny", "tiny64x64", "fwd_dpt_tiny_64x64.npz", 2, 64, 64),
    ("dpt_tiny", "tiny", "tiny96x64", "fwd_dpt_tiny_96x64.npz", 2, 96, 64),
    ("dpt_base", "base", "base128", "fwd_dpt_base_128.npz", 1, 128, 128),
    ("dpt_large", "large", "large128", "fwd_dpt_large_128.npz", 1, 128, 128),
]


@pytest.mark.parametrize("mode", ["collapsed", "auto"])
@pytest.mark.parametrize("cfg,tag,img,fname,B,H,W", FULL)
def test_collapsed_forward_fp32_matches_reference_golden(golden_dir, cfg, tag, img, fname, B, H, W, mode):
    """the four full-map fixtures made by the reference's own modules, at the north star's 1e-4"""
    g = np.load(os.path.join(golden_dir, fname))
    net, _ = _net(cfg, tag, mode=mode)
    x = torch.from_numpy(uniform01(f"img:{img}", (B, 3, H, W))).cuda()
    with torch.no_grad():
        out = net(images=x)
        out2 = net.get_prediction(x)
    assert out["sdf_maps"].shape == (B, 1, H, W) and out["sdf_maps"].dtype == torch.float32
    e_c = np.abs(out["center_fields"].cpu().numpy() - g["center_fields"]).max()
    e_s = np.abs(out["sdf_maps"].cpu().numpy() - g["sdf_maps"]).max()
    print(f"{fname} [{mode}]: max |center - ref| = {e_c:.2e}, max |sdf - ref| = {e_s:.2e}")
    assert e_c <= 1e-4 and e_s <= 1e-4
    assert torch.equal(out["center_fields"], out2["center_fields"]) and torch.equal(out["sdf_maps"], out2["sdf_maps"])


def test_collapsed_forward_fp32_benchmark_size_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "fwd_dpt_base_384_sampled.npz"))
    net, _ = _net("dpt_base", tag="base", size=384)
    x = torch.from_numpy(synth.blob_images(1, 384, 384, seed=11)).cuda()
    with torch.no_grad():
        out = net(images=x)
    idx = g["sample_idx"]
    cen, sdf = out["center_fields"][0].cpu(), out["sdf_maps"][0].cpu()
    np.testing.assert_allclose(cen.reshape(2, -1)[:, idx].numpy(), g["center_samples"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(sdf.reshape(1, -1)[:, idx].numpy(), g["sdf_samples"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(sdf.mean(dim=(1, 2)).numpy(), g["sdf_mean"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(sdf.abs().amax(dim=(1, 2)).numpy(), g["sdf_absmax"], atol=1e-4, rtol=0)


def test_collapsed_forward_fp32_cfg1_shape_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "fwd_dpt_small_224_sampled.npz"))
    net, _ = _net("dpt_small", tag="dpt_small", size=224)
    x = torch.from_numpy(synth.blob_images(2, 224, 224, seed=12)).cuda()
    with torch.no_grad():
        out = net(images=x)
    idx = g["sample_idx"]
    sdf, cen = out["sdf_maps"].cpu(), out["center_fields"].cpu()
    np.testing.assert_allclose(sdf.reshape(2, 1, -1)[:, :, idx].numpy().reshape(g["sdf_samples"].shape), g["sdf_samples"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(cen.reshape(2, 2, -1)[:, :, idx].numpy().reshape(g["center_samples"].shape), g["center_samples"], atol=1e-4, rtol=0)


@pytest.mark.parametrize("tag", sorted(pc.E2E))
def test_collapsed_forward_to_peaks_matches_reference_chain(tag):
    """north star: 'bit-exact for the argmax peak indices feeding object_reasoning' with the collapsed head in the chain --
    the same statement as tests/test_parity_r2_gpu.py::test_hip_forward_to_peaks_matches_reference_chain makes for the factored one"""
    from unmore_amd import reasoning
    g = pc.load()
    cfg_name, wtag = pc.E2E[tag]
    shift, scale = g[f"{tag}_meta_shift_scale"]
    sd = pc.edited_state_dict(orc.state_dict_spec(orc.CONFIGS[cfg_name]), wtag, shift, scale)
    net, _ = _net(cfg_name, sd=sd)
    x = pc.e2e_images(tag).cuda()
    with torch.no_grad():
        out = net.get_prediction(x)
    sdf, cen = out["sdf_maps"].squeeze(1), out["center_fields"]
    idx = g[f"{tag}_sample_idx"]
    e1 = np.abs(sdf.reshape(8, -1)[:, idx].cpu().numpy() - g[f"{tag}_sdf_samples"]).max()
    e2 = np.abs(cen.reshape(8, 2, -1)[:, :, idx].cpu().numpy() - g[f"{tag}_center_samples"]).max()
    assert max(e1, e2) < 1e-4, (e1, e2)
    mx, am = reasoning.center_peaks(sdf, cen)
    mx, am = mx.cpu().numpy(), am.cpu().numpy()
    report = []
    n = pc.check_peaks_against_fixture(g, tag, mx, am, field_err=float(max(e1, e2, 1e-6)), report=report)
    n_equal = int((am == g[f"{tag}_argmax"]).sum())
    print(f"{tag} [collapsed]: field err {max(e1, e2):.2e}; argmax equal on {n_equal}/8 maps ({n} certified); differences: {report or 'none'}")
    assert n_equal == 8 and not report, report


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 3e-2)])
@pytest.mark.parametrize("act,bg", [("tanh", True), (None, True), ("sine", True)])
def test_collapsed_equals_factored_inference_all_linear_variants(dtype, tol, act, bg):
    """the three activation-free variants of the head (objectness_net.py:119-142); the ReLU variants never collapse"""
    args = Namespace(use_bg_sdf=bg, sdf_activation=act)
    x = torch.from_numpy(uniform01("img:cvf", (3, 3, 64, 96))).cuda()
    outs = {}
    for mode in ("factored", "collapsed"):
        net, sd = _net("dpt_tiny", "tiny", dtype=dtype, args=args, mode=mode, size=64)
        with torch.no_grad():
            outs[mode] = net.get_prediction(x)
    d = (outs["factored"]["sdf_maps"] - outs["collapsed"]["sdf_maps"]).abs().max().item()
    print(f"{act} {dtype}: max |factored - collapsed| = {d:.2e}")
    assert d <= tol
    assert torch.equal(outs["factored"]["center_fields"], outs["collapsed"]["center_fields"])
    if dtype == torch.float32:
        ref = orc.forward({k: v.double() for k, v in sd.items()}, x.cpu().double(), orc.CONFIGS["dpt_tiny"], use_bg_sdf=bg, sdf_activation=act)
        assert (outs["collapsed"]["sdf_maps"].cpu().double() - ref["sdf_maps"]).abs().max().item() <= 1e-4


def test_relu_variants_never_collapse_and_gemm_backward_training_stays_factored():
    for args in (Namespace(use_bg_sdf=True, sdf_activation="relu"), Namespace(use_bg_sdf=False, sdf_activation="tanh")):
        net, _ = _net("dpt_tiny", "tiny", args=args, mode="collapsed", size=64)
        eng = net._engine()
        assert not eng._collapse(eng.sdf_layout, False) and not eng._collapse(eng.center_layout, False)
    # the default: collapsed at inference and -- since round 6 -- in training when the backward is the algebraic one (it reads the
    # head's output only; tests/test_collapsed_train_gpu.py).  With the layer-by-layer GEMM backward, which reads the four
    # convolutions' activations, training under autograd runs them: bit-identical to 'factored'
    net, _ = _net("dpt_tiny", "tiny", mode="auto", size=64)
    eng = net._engine()
    assert eng._collapse(eng.sdf_layout, save=False) and eng._collapse(eng.sdf_layout, save=True)
    assert not eng._collapse(eng.center_layout, save=False)
    net.set_linear_head_backward("gemm")
    eng = net._engine()
    assert eng._collapse(eng.sdf_layout, save=False) and not eng._collapse(eng.sdf_layout, save=True)
    x = torch.from_numpy(uniform01("img:tf", (2, 3, 64, 64))).cuda()
    net_f, _ = _net("dpt_tiny", "tiny", mode="factored", size=64)
    net_f.set_linear_head_backward("gemm")
    net.train()
    net_f.train()
    o_a, o_f = net(images=x), net_f(images=x)
    assert torch.equal(o_a["sdf_maps"], o_f["sdf_maps"]) and torch.equal(o_a["center_fields"], o_f["center_fields"])


def test_collapsed_weights_follow_parameter_changes():
    """the collapsed weights are cached per version of the head's eight tensors: an in-place change of ANY of them (here the third
    layer's bias and the second layer's weight, neither of which keys another pack) is seen by the next call, eager and replayed"""
    x = torch.from_numpy(uniform01("img:cw", (2, 3, 64, 64))).cuda()
    for gmode in ("off", "on"):
        net_c, _ = _net("dpt_tiny", "tiny", mode="collapsed", size=64)
        net_f, _ = _net("dpt_tiny", "tiny", mode="factored", size=64)
        net_c.set_graph_mode(gmode)
        net_f.set_graph_mode("off")
        with torch.no_grad():
            for it in range(4):
                a, b = net_c.get_prediction(x)["sdf_maps"], net_f.get_prediction(x)["sdf_maps"]
                assert (a - b).abs().max().item() <= 2e-5, (gmode, it)
            before = a.clone()
            for net in (net_c, net_f):
                net.sdf_prediction_head[2].bias.add_(0.05)
                net.sdf_prediction_head[1].weight.mul_(1.1)
            for it in range(4):
                a, b = net_c.get_prediction(x)["sdf_maps"], net_f.get_prediction(x)["sdf_maps"]
                assert (a - b).abs().max().item() <= 2e-5, (gmode, it)
            assert (a - before).abs().max().item() > 1e-3


def test_gather9_kernel_against_direct_sum():
    from unmore_amd import _lib as L
    from unmore_amd import ops
    torch.manual_seed(0)
    B, H, W = 2, 5, 7
    taps = torch.randn(B, H, W, 16, device="cuda")
    tb = torch.randn(10, device="cuda")
    out = ops.linear_head_gather9(taps, tb, L.ACT_TANH).cpu().double()
    t, b = taps.cpu().double(), tb.cpu().double()
    ref = torch.zeros(B, 1, H, W, dtype=torch.float64)
    for y in range(H):
        for x_ in range(W):
            s = torch.full((B,), float(b[9]), dtype=torch.float64)
            for k in range(9):
                yy, xx = y + k // 3 - 1, x_ + k % 3 - 1
                if 0 <= yy < H and 0 <= xx < W:
                    s = s + t[:, yy, xx, k] + b[k]
            ref[:, 0, y, x_] = torch.tanh(s)
    assert (out - ref).abs().max().item() < 1e-6


@pytest.mark.parametrize("cfg,tag,img,fname,B,H,W", FULL)
def test_three_term_product_mode_on_the_reference_fixtures(golden_dir, cfg, tag, img, fname, B, H, W):
    """The opt-in three-term plane products (umr_set_f32_mode(UMR_F32_X3_FAST): products to 2^-16 instead of 2^-22, half the matrix
    work of the fp32 mode's heads; bench.py reports it beside the cfg5 headline as alt_fp32_3term) against the same reference-made
    fixtures: inside the 1e-4 contract, but with a fraction of the default mode's margin -- the measured errors are printed next to the
    six-term mode's, and that margin is why it stays opt-in."""
    from unmore_amd import ops
    g = np.load(os.path.join(golden_dir, fname))
    net, _ = _net(cfg, tag, mode="auto")
    x = torch.from_numpy(uniform01(f"img:{img}", (B, 3, H, W))).cuda()
    errs = {}
    try:
        for mode in ("x3", "x3_fast"):
            ops.set_f32_mode(mode)
            with torch.no_grad():
                out = net.get_prediction(x)
            errs[mode] = (float(np.abs(out["center_fields"].cpu().numpy() - g["center_fields"]).max()),
                          float(np.abs(out["sdf_maps"].cpu().numpy() - g["sdf_maps"]).max()))
    finally:
        ops.set_f32_mode("x3")
    print(f"{fname}: max |map - reference|  six-term {errs['x3'][0]:.2e} / {errs['x3'][1]:.2e}   three-term {errs['x3_fast'][0]:.2e} / {errs['x3_fast'][1]:.2e}")
    assert max(errs["x3"]) <= 1e-4 and max(errs["x3_fast"]) <= 1e-4
    assert max(errs["x3"]) <= 2e-5          # the default mode keeps at least 5x of margin on every fixture
