"""oracle/mask_parity.py on the CPU: imposing the oracle's OWN ReLU decisions must change nothing (outputs and gradients
bit-identical up to the x*mask vs max(x,0) identity), and imposing another precision's decisions must report the flips."""
import torch

from oracle import mask_parity
from oracle import objectness_oracle as orc
from unmore_amd import synth
from unmore_amd.hashrng import hash_init


def _record_masks(sd, img, cfg):
    masks = {}

    def hook(x, site):
        masks[site] = (x.detach() > 0)
        return torch.relu(x)

    with orc.relu_hook(hook):
        out = orc.forward(sd, img, cfg)
    return out, masks


def test_own_decisions_reproduce_forward_and_gradients():
    cfg = orc.CONFIGS["dpt_tiny"]
    sd = {k: torch.from_numpy(hash_init(k, s, "tiny")).double().requires_grad_(True) for k, s in orc.state_dict_spec(cfg).items()}
    img = torch.from_numpy(synth.blob_images(1, 64, 64, seed=3)).double()
    out, masks = _record_masks(sd, img, cfg)
    # 2 RCUs x 2 sites x 4 refinenets minus refinenet4.resConfUnit1 (never run, models.py:85) + 3 centre-head sites
    assert len(masks) == 14 + 3 and not any("refinenet4.resConfUnit1" in k for k in masks)
    assert not any(k.startswith("sdf_prediction_head") for k in masks)   # the tanh head has no ReLU
    out_m, flips = mask_parity.masked_forward(sd, img, cfg, masks)
    assert sum(flips.values()) == 0
    names = [n for n in sd]
    cot = [torch.ones_like(out["center_fields"]), torch.ones_like(out["sdf_maps"])]
    g0 = torch.autograd.grad([out["center_fields"], out["sdf_maps"]], [sd[n] for n in names], grad_outputs=cot, allow_unused=True)
    g1 = torch.autograd.grad([out_m["center_fields"], out_m["sdf_maps"]], [sd[n] for n in names], grad_outputs=cot, allow_unused=True)
    for k in ("center_fields", "sdf_maps"):
        assert torch.equal(out[k], out_m[k])
    for n, a, b in zip(names, g0, g1):
        assert (a is None) == (b is None), n
        if a is not None:
            assert torch.equal(a, b), n


def test_foreign_decisions_are_counted():
    cfg = orc.CONFIGS["dpt_tiny"]
    spec = orc.state_dict_spec(cfg)
    sd32 = {k: torch.from_numpy(hash_init(k, s, "tiny")) for k, s in spec.items()}
    img = torch.from_numpy(synth.blob_images(1, 64, 64, seed=3))
    _, masks32 = _record_masks(sd32, img, cfg)
    some = "center_field_prediction_head.relu2"   # the last site: nothing downstream of it is a ReLU
    sd64 = {k: v.double() for k, v in sd32.items()}
    _, flips = mask_parity.masked_forward(sd64, img.double(), cfg, masks32)
    n_all = sum(m.numel() for m in masks32.values())
    assert sum(flips.values()) < 1e-3 * n_all     # fp32 and float64 disagree only within rounding of zero
    base = flips[some]
    masks32[some] = ~masks32[some]
    _, flips = mask_parity.masked_forward(sd64, img.double(), cfg, masks32)
    assert flips[some] == masks32[some].numel() - base
