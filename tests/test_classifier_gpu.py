"""Existence classifier on the MI355X: kernels vs PyTorch references of the same op, and the drop-in
`Binary_Classifier` vs the CPU oracle / the fixtures made by the reference's own class."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import classifier_oracle as CO
from unmore_amd.hashrng import uniform, uniform01

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _dev():
    assert torch.cuda.is_available(), "GPU test run without a GPU"
    return torch.device("cuda:0")


def _rnd(shape, seed, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_im2col_maxpool_bnfold(dtype):
    from unmore_amd import ops
    dev = _dev()
    x = _rnd((2, 3, 37, 50), 1)
    cols, Ho, Wo = ops.im2col_nchw(x.to(dev), 7, 7, 2, 3, 152, dtype)
    ref = F.unfold(x, kernel_size=7, padding=3, stride=2).transpose(1, 2).reshape(-1, 147)   # K order (c, ky, kx)
    assert (Ho, Wo) == (19, 25) and cols.shape == (2 * 19 * 25, 152)
    tol = dict(atol=0, rtol=0) if dtype == torch.float32 else dict(atol=2e-2, rtol=2e-2)
    torch.testing.assert_close(cols[:, :147].float().cpu(), ref, **tol)
    assert float(cols[:, 147:].float().abs().max()) == 0.0
    # max-pool, odd sizes, NHWC
    h = _rnd((2, 13, 18, 8), 2)
    y = ops.maxpool3x3s2(h.to(dev).to(dtype))
    refp = F.max_pool2d(h.to(dtype).float().permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1)
    torch.testing.assert_close(y.float().cpu(), refp, atol=0, rtol=0)
    # BatchNorm folding
    w = _rnd((16, 27), 3)
    g, b, m, v = _rnd((16,), 4).abs() + 0.5, _rnd((16,), 5), _rnd((16,), 6), _rnd((16,), 7).abs() + 0.3
    wf, bf = ops.bn_fold(w.to(dev), g.to(dev), b.to(dev), m.to(dev), v.to(dev), 1e-5, 32, dtype)
    s = g / torch.sqrt(v + 1e-5)
    torch.testing.assert_close(wf[:, :27].float().cpu(), w * s[:, None], **(dict(atol=1e-6, rtol=1e-6) if dtype == torch.float32 else dict(atol=2e-2, rtol=2e-2)))
    assert float(wf[:, 27:].float().abs().max()) == 0.0
    torch.testing.assert_close(bf.cpu(), b - m * s, atol=1e-6, rtol=1e-6)


def _net(dtype):
    from unmore_amd.binary_classifier import Binary_Classifier
    net = Binary_Classifier(device="cuda:0", image_size=128, args=None, compute_dtype=dtype)
    net.load_state_dict(CO.hash_state("clf", uniform), strict=True)
    return net.to(_dev()).eval()


@pytest.mark.parametrize("B,S", [(2, 64), (3, 128)])
def test_classifier_fp32_matches_reference_fixture(B, S):
    net = _net(torch.float32)
    x = torch.from_numpy(uniform01(f"img:clf{S}", (B, 3, S, S))).to(_dev())
    with torch.no_grad():
        y = net(x)
    assert y.shape == (B, 1) and y.dtype == torch.float32
    want = np.load(os.path.join(GOLD, f"clf_fwd_{S}.npz"))["prob"]
    np.testing.assert_allclose(y.cpu().numpy(), want, rtol=0, atol=1e-4)   # north_star tolerance: fp32 within 1e-4


def test_classifier_odd_size_batch_and_bf16():
    """non-square, non-multiple-of-32 crops (every stride-2 stage sees odd extents) and a varied batch; logits compared
    through the inverse sigmoid so that the check is not flattened by the output non-linearity"""
    sd = CO.hash_state("clf", uniform)
    x = torch.from_numpy(uniform01("img:clf_odd", (5, 3, 90, 70)))
    x = x * torch.linspace(0.2, 3.0, 5).view(5, 1, 1, 1)       # spread the operating points
    want = CO.forward({k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}, x.double())
    net = _net(torch.float32)
    with torch.no_grad():
        y32 = net(x.to(_dev())).double().cpu()
    torch.testing.assert_close(y32, want, atol=1e-4, rtol=0)
    torch.testing.assert_close(torch.logit(y32), torch.logit(want), atol=2e-3, rtol=0)
    net.set_compute_dtype(torch.bfloat16)
    with torch.no_grad():
        y16 = net(x.to(_dev())).double().cpu()
    torch.testing.assert_close(y16, want, atol=3e-2, rtol=0)


def test_classifier_refuses_training_mode():
    net = _net(torch.float32)
    x = torch.zeros(1, 3, 64, 64, device=_dev())
    net.train()
    with pytest.raises(NotImplementedError):
        net(x)
    net.eval()
    with pytest.raises(NotImplementedError):   # gradients enabled + trainable parameters: not the inference path
        net(x)
    with torch.no_grad():
        assert net(x).shape == (1, 1)


def test_existence_checking_matches_oracle():
    """object_reasoning.py:491-523: crop + Resize(128) + classifier over > 1 batch of 128 proposals"""
    from oracle import objectness_oracle as O
    from unmore_amd.reasoning import existence_checking
    dev = _dev()
    image = torch.from_numpy(uniform01("img:clf_scene", (3, 120, 160)))
    g = torch.Generator(device="cpu").manual_seed(3)
    n = 131
    x1 = torch.rand(n, generator=g) * 100
    y1 = torch.rand(n, generator=g) * 70
    boxes = torch.stack([x1, y1, x1 + 20 + torch.rand(n, generator=g) * 39.5, y1 + 20 + torch.rand(n, generator=g) * 29.5], 1)
    net = _net(torch.float32)
    got = existence_checking(net, image.to(dev), boxes)["existence_scores"]
    assert got.shape == (n,) and got.device.type == "cpu"
    sd = CO.hash_state("clf", uniform)
    want = CO.existence_scores(sd, image, boxes, O.crop_resize)
    torch.testing.assert_close(got, want, atol=1e-4, rtol=0)


def test_get_prediction_with_proposals_is_the_composition_of_its_parts():
    """object_scoring.py:111-153: crops -> objectness net + existence classifier in batches of 50; the helper must return
    exactly what the three pieces give when chained by hand."""
    from argparse import Namespace
    from unmore_amd import reasoning
    from unmore_amd.hashrng import hash_init
    from unmore_amd.objectness_net import ObjectnessNet
    dev = _dev()
    obj = ObjectnessNet("cuda:0", 128, "dpt_tiny", Namespace(use_bg_sdf=True, sdf_activation="tanh"))
    obj.load_state_dict({k: torch.from_numpy(hash_init(k, tuple(v.shape), "tiny")) for k, v in obj.state_dict().items()})
    obj = obj.to("cuda:0").eval()
    clf = _net(torch.float32)
    image = torch.from_numpy(uniform01("img:clf_scene", (3, 120, 160))).to(dev)
    g = torch.Generator(device="cpu").manual_seed(5)
    n = 57    # > one batch of 50
    x1 = torch.rand(n, generator=g) * 100
    y1 = torch.rand(n, generator=g) * 70
    boxes = torch.stack([x1, y1, x1 + 20 + torch.rand(n, generator=g) * 39.5, y1 + 20 + torch.rand(n, generator=g) * 29.5], 1)
    boxes[0] = torch.tensor([0.0, 0.0, 160.0, 120.0])
    got = reasoning.get_prediction_with_proposals(obj, clf, image, boxes.tolist())
    assert got["pred_boundary_fields"].shape == (n, 128, 128) and got["pred_center_fields"].shape == (n, 2, 128, 128)
    assert got["pred_existence_scores"].shape == (n,) and bool(got["on_edge_flags"][0].all())
    crops, _ = reasoning.crop_resize(image, boxes, 128)
    with torch.no_grad():
        for lo in (0, 50):
            pred = obj.get_prediction(crops[lo:lo + 50])
            assert torch.equal(got["pred_boundary_fields"][lo:lo + 50], pred["sdf_maps"].squeeze(1))
            assert torch.equal(got["pred_center_fields"][lo:lo + 50], pred["center_fields"])
            assert torch.equal(got["pred_existence_scores"][lo:lo + 50], clf(crops[lo:lo + 50]).squeeze(1))
