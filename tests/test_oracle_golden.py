"""Pin the CPU oracle against fixtures produced by the reference's own modules
(tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import objectness_oracle as orc
from unmore_amd.hashrng import hash_init, uniform01


def _sd(cfg_name, tag):
    spec = orc.state_dict_spec(orc.CONFIGS[cfg_name])
    return {k: torch.from_numpy(hash_init(k, s, tag)) for k, s in spec.items()}


def _schema(path):
    out = []
    for line in open(path):
        parts = line.split()
        out.append((parts[0], tuple(int(x) for x in parts[1:])))
    return out


@pytest.mark.parametrize("cfg,fname", [("dpt_large", "schema_dpt_large.txt"),
                                       ("dpt_base", "schema_dpt_base.txt"),
                                       ("dpt_tiny", "schema_dpt_tiny.txt")])
def test_state_dict_schema_matches_reference(golden_dir, cfg, fname):
    ref = _schema(os.path.join(golden_dir, fname))
    mine = [(k, tuple(v)) for k, v in orc.state_dict_spec(orc.CONFIGS[cfg]).items()]
    assert mine == ref
    if cfg == "dpt_large":
        assert len(mine) == 378
        assert sum(int(np.prod(s)) for _, s in mine) == 349_759_979


@pytest.mark.parametrize("cfg,tag,img,fname,B,H,W", [
    ("dpt_tiny", "tiny", "tiny64x64", "fwd_dpt_tiny_64x64.npz", 2, 64, 64),
    ("dpt_tiny", "tiny", "tiny96x64", "fwd_dpt_tiny_96x64.npz", 2, 96, 64),
    ("dpt_base", "base", "base128", "fwd_dpt_base_128.npz", 1, 128, 128),
    ("dpt_large", "large", "large128", "fwd_dpt_large_128.npz", 1, 128, 128),
])
def test_oracle_forward_matches_reference(golden_dir, cfg, tag, img, fname, B, H, W):
    g = np.load(os.path.join(golden_dir, fname))
    sd = _sd(cfg, tag)
    x = torch.from_numpy(uniform01(f"img:{img}", (B, 3, H, W)))
    inter = {}
    with torch.no_grad():
        out = orc.forward(sd, x, orc.CONFIGS[cfg], inter=inter)
    # fp32 CPU vs fp32 CPU, different op order only in attention (SDPA vs explicit)
    np.testing.assert_allclose(out["center_fields"].numpy(), g["center_fields"], atol=2e-5, rtol=1e-4)
    np.testing.assert_allclose(out["sdf_maps"].numpy(), g["sdf_maps"], atol=2e-5, rtol=1e-4)
    if "inter_output_conv" in g:
        feat = inter["feat"].numpy()[:, ::8, ::2, ::2]
        np.testing.assert_allclose(feat, g["inter_output_conv"], atol=2e-5, rtol=1e-4)
        p4 = inter["path_4"].numpy()
        ref = g["inter_refinenet4"]
        if ref.shape != p4.shape:
            p4 = p4[:, ::8, ::2, ::2]
        np.testing.assert_allclose(p4, ref, atol=2e-5, rtol=1e-4)


def test_nograd_list_matches_reference(golden_dir):
    """parameters that never receive a gradient."""
    ref = [l.strip() for l in open(os.path.join(golden_dir, "nograd_dpt_large.txt")) if l.strip()]
    cfg = orc.CONFIGS["dpt_tiny"]
    sd = {k: v.requires_grad_(True) for k, v in _sd("dpt_tiny", "tiny").items()}
    x = torch.from_numpy(uniform01("img:tiny64x64", (2, 3, 64, 64)))
    out = orc.forward(sd, x, cfg)
    (out["center_fields"].mean() + out["sdf_maps"].mean()).backward()
    mine = sorted(k for k, v in sd.items() if v.grad is None)
    assert mine == sorted(ref)


def test_image_gradients_and_loss_autograd():
    torch.manual_seed(0)
    ps = torch.tanh(torch.randn(2, 1, 8, 6)).requires_grad_(True)
    pc = torch.randn(2, 2, 8, 6, requires_grad=True)
    gc, gs = torch.randn(2, 2, 8, 6), torch.tanh(torch.randn(2, 1, 8, 6))
    sal = (torch.rand(2, 1, 8, 6) > 0.5).float()
    dy, dx = orc.image_gradients(gs)
    assert torch.all(dy[..., -1, :] == 0) and torch.all(dx[..., :, -1] == 0)
    assert torch.allclose(dy[..., :-1, :], gs[..., 1:, :] - gs[..., :-1, :])
    total, terms = orc.loss_terms({"center_fields": pc, "sdf_maps": ps}, gc, gs, sal)
    assert len(terms) == 4
    total.backward()
    assert ps.grad.abs().sum() > 0 and pc.grad.abs().sum() > 0
    assert torch.allclose(terms[0], ((pc - gc) ** 2).mean())
