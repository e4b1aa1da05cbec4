"""CPU ORACLE SUPPORT -- TEST INFRASTRUCTURE ONLY (imported by tests/ and __graft_entry__.smoke(), never by unmore_amd/).

Gradient parity on the same linear piece.  The network is piecewise linear in its ReLUs (blocks.py:290-313 RCUs, the
centre head objectness_net.py:109-117); where a pre-activation is within rounding of zero, fp32 and float64 (and two fp32
implementations) decide the mask differently and single gradient elements move by ~1e-3 -- noise of the function's kinks,
not of the backward kernels.  `hip_relu_masks` reads the decisions the HIP path actually took (its saved post-ReLU
activations) and `masked_forward` runs the float64 oracle with exactly those decisions imposed, so that the oracle's
vector-Jacobian product is the exact gradient of what the HIP backward differentiates."""
import torch

from . import objectness_oracle as orc


def _nchw_mask(t, B, H, W):
    if hasattr(t, "mask_source"):   # the fp32 mode saves activations as bf16 planes: the leading plane has the value's sign
        t = t.mask_source()
    return (t.reshape(B, H, W, -1) > 0).permute(0, 3, 1, 2).cpu()


def hip_relu_masks(S, head_layouts):
    """S: the saved-activation dict of unmore_amd.engine.Engine.forward(save=True), BEFORE backward consumes it.
    head_layouts: (centre, sdf) layout dicts (Engine.center_layout / .sdf_layout; `relu` says whether the head has ReLUs).
    Returns {site name (oracle.objectness_oracle ReLU sites): bool NCHW mask on the CPU}."""
    B, H, W = S["B"], S["H"], S["W"]
    masks = {}
    for k, fs in S["fus"].items():
        r = f"backbone.scratch.refinenet{k}."
        hh, ww = fs["in_hw"]
        masks[r + "resConfUnit2.relu_in"] = _nchw_mask(fs["s_relu"], B, hh, ww)
        masks[r + "resConfUnit2.relu_mid"] = _nchw_mask(fs["t2"], B, hh, ww)
        if "t1" in fs:
            masks[r + "resConfUnit1.relu_in"] = _nchw_mask(fs["x1_relu"], B, hh, ww)
            masks[r + "resConfUnit1.relu_mid"] = _nchw_mask(fs["t1"], B, hh, ww)
    for hs, lay, name in zip(S["heads"], head_layouts, ("center_field_prediction_head", "sdf_prediction_head")):
        if lay["relu"]:
            for i, key in enumerate(("h1", "h2", "h3")):
                masks[f"{name}.relu{i}"] = _nchw_mask(hs[key], B, H, W)
    return masks


class FlipCounts(dict):
    """{site: number of decisions where the oracle's own x > 0 differs from the imposed mask}, plus
    .sites  = total number of decisions, and
    .margin = the largest |pre-activation| among the flipped elements, in units of its tensor's rms: a flip is legitimate only where
              the pre-activation is within rounding of zero -- a wrong mask far from zero would show up here as O(1)."""
    sites = 0
    margin = 0.0


def masked_forward(sd, images, cfg, masks, **kw):
    """orc.forward with the given ReLU decisions imposed (relu(x) := x * mask); returns (out dict, FlipCounts)."""
    flips = FlipCounts()

    def hook(x, site):
        m = masks[site]
        assert m.shape == x.shape, (site, tuple(m.shape), tuple(x.shape))
        xd = x.detach()
        diff = (xd > 0) != m
        flips[site] = int(diff.sum())
        flips.sites += m.numel()
        if flips[site]:
            rms = xd.double().pow(2).mean().sqrt().item() + 1e-300
            flips.margin = max(flips.margin, xd[diff].abs().max().item() / rms)
        return x * m.to(x.dtype)

    with orc.relu_hook(hook):
        out = orc.forward(sd, images, cfg, **kw)
    assert set(flips) == set(masks), (sorted(set(masks) ^ set(flips)))
    return out, flips


def assert_flips_are_rounding(flips, max_share=1e-5, max_margin=1e-4):
    """the hypothesis the masked comparison rests on, asserted: only a vanishing share of the decisions differs from float64's own
    (measured 5 of 17.5 M, 40 of 69.9 M), and every one of them sits within rounding of the kink (fp32 rounding of a
    pre-activation of rms 1 is ~1e-6; 1e-4 leaves two orders of margin and still rejects a genuinely wrong mask)"""
    n = sum(flips.values())
    assert n <= max(1.0, max_share * flips.sites), (n, flips.sites)
    assert flips.margin <= max_margin, flips.margin
