"""Pin the CPU oracle against the round-2 fixtures made from the reference's own code (tests/golden/make_golden_r2.py):
(ii) the 4-term loss block train_objectness_net.py:215-254 exec'd from the reference file, (iii) a benchmark-size forward
(ViT-B/16 wiring, 384x384) through the reference modules, (iv) the peak-picking chain of object_reasoning.py:525-557 /
utils/misc.py:10-20 / :139-174 run on reference-net outputs.  CPU only."""
import os

import numpy as np
import pytest
import torch

import peaks_common as pc
from oracle import objectness_oracle as orc
from unmore_amd import synth
from unmore_amd.hashrng import hash_init, uniform, uniform01


def loss_inputs():
    B, H, W = 2, 12, 10
    return (torch.from_numpy(uniform("loss:pc", (B, 2, H, W), -1.2, 1.2)), torch.from_numpy(uniform("loss:ps", (B, 1, H, W), -0.98, 0.98)),
            torch.from_numpy(uniform("loss:gc", (B, 2, H, W), -1.0, 1.0)), torch.from_numpy(uniform("loss:gs", (B, 1, H, W), -1.0, 1.0)),
            torch.from_numpy((uniform01("loss:sal", (B, 1, H, W)) > 0.5).astype(np.float32)))


LOSS_COMBOS = [(cl, sl, ug, ub) for cl in ("l2", "l1") for sl in ("l1", "l2") for ug in (0, 1) for ub in (0, 1)]


@pytest.mark.parametrize("cl,sl,ug,ub", LOSS_COMBOS)
def test_loss_oracle_matches_reference_block(golden_dir, cl, sl, ug, ub):
    g = np.load(os.path.join(golden_dir, "loss_terms.npz"))
    pcn, ps, gc, gs, sal = loss_inputs()
    pcn.requires_grad_(True)
    ps.requires_grad_(True)
    total, _ = orc.loss_terms({"center_fields": pcn, "sdf_maps": ps}, gc, gs, sal, cl, sl, bool(ug), bool(ub))
    total.backward()
    key = f"{cl}_{sl}_g{ug}_b{ub}"
    assert abs(total.item() - float(g[key + "_loss"])) <= 1e-6
    np.testing.assert_allclose(pcn.grad.numpy(), g[key + "_dpc"], atol=1e-8, rtol=1e-5)
    np.testing.assert_allclose(ps.grad.numpy(), g[key + "_dps"], atol=1e-8, rtol=1e-5)


def _oracle_chain(sdf, cen):
    score, amax, arg = orc.peak_pick(sdf, cen)
    return score, amax.numpy(), arg.numpy()


@pytest.mark.parametrize("tag", sorted(pc.SYN) + sorted(pc.E2E))
def test_peak_oracle_matches_reference_functions(tag):
    """oracle peak chain / box deltas on the fixture's exact input maps == the reference functions' results, bit for bit on
    the integer side (eroded mask, argmax) and to float64 round-off on the scores"""
    g = pc.load()
    if tag in pc.SYN:
        B, H, W, seed = pc.SYN[tag]
        sdf, cen = (torch.from_numpy(a) for a in synth.object_like_fields(B, H, W, seed))
    else:
        sdf, cen = torch.from_numpy(g[f"{tag}_sdf_maps"]), torch.from_numpy(g[f"{tag}_center_fields"])
        B, H, W = sdf.shape
    score, amax, arg = _oracle_chain(sdf, cen)
    sel = slice(0, B)
    assert (g[f"{tag}_amax"][sel] > 0).any(), "vacuous fixture"
    np.testing.assert_array_equal(arg, g[f"{tag}_argmax"][sel])
    np.testing.assert_allclose(amax, g[f"{tag}_amax"][sel], atol=1e-13, rtol=0)
    np.testing.assert_array_equal((score != 0).reshape(B, -1).sum(1).numpy(), g[f"{tag}_score_support"][sel])
    np.testing.assert_allclose(score[:, H // 2, :].numpy(), g[f"{tag}_score_at_rows"][sel], atol=1e-13, rtol=0)
    # the eroded mask itself
    union = torch.where((torch.where(torch.norm(cen, dim=1) > 0.5, 1, 0) + torch.where(torch.sigmoid(sdf) > 0.5, 1, 0)) > 0, 1, 0)
    er = orc.batch_erode(union, 9, 3).numpy().astype(bool).reshape(B, -1)
    np.testing.assert_array_equal(er, pc.eroded_mask(g, tag, B, H * W)[sel])
    d = torch.stack(orc.update_bbox_with_boundary_fields(sdf), 1).numpy()
    np.testing.assert_allclose(d, g[f"{tag}_deltas"][sel], atol=1e-5, rtol=1e-5)
    # unravel (object_reasoning.py:198-204, :550): (idx // W, idx % W)
    yx = g[f"{tag}_peak_yx"][sel]
    for b in range(B):
        if yx[b, 0] >= 0:
            assert (int(arg[b]) // W, int(arg[b]) % W) == (int(yx[b, 0]), int(yx[b, 1]))


@pytest.mark.parametrize("tag", sorted(pc.SYN) + sorted(pc.E2E))
def test_oracle_peak_certificate_matches_the_fixture(tag):
    """oracle.peak_certificate (used by bench.py --workload cfg5 to say which peak-index mismatches would be real errors) against
    the certificate the fixture generator computed with the reference's own erode / score functions: same certified set"""
    g = pc.load()
    if tag in pc.SYN:
        B, H, W, seed = pc.SYN[tag]
        sdf, cen = (torch.from_numpy(a) for a in synth.object_like_fields(B, H, W, seed))
    else:
        sdf, cen = torch.from_numpy(g[f"{tag}_sdf_maps"]), torch.from_numpy(g[f"{tag}_center_fields"])
        B = sdf.shape[0]
    amax, arg, cert = orc.peak_certificate(sdf, cen, float(g["meta_cert_eps"]))
    np.testing.assert_array_equal(arg.numpy(), g[f"{tag}_argmax"][:B])
    np.testing.assert_array_equal(cert.numpy(), g[f"{tag}_argmax_certified"][:B].astype(bool))


@pytest.mark.parametrize("tag", sorted(pc.E2E))
def test_oracle_forward_to_peaks_matches_reference(tag):
    """oracle forward (edited hash weights, blob images) -> oracle peaks == reference forward -> reference peaks"""
    g = pc.load()
    cfg_name, wtag = pc.E2E[tag]
    shift, scale = g[f"{tag}_meta_shift_scale"]
    cfg = orc.CONFIGS[cfg_name]
    sd = pc.edited_state_dict(orc.state_dict_spec(cfg), wtag, shift, scale)
    x = pc.e2e_images(tag)
    with torch.no_grad():
        out = orc.forward(sd, x, cfg)
    sdf, cen = out["sdf_maps"].squeeze(1), out["center_fields"]
    idx = g[f"{tag}_sample_idx"]
    e1 = np.abs(sdf.reshape(8, -1)[:, idx].numpy() - g[f"{tag}_sdf_samples"]).max()
    e2 = np.abs(cen.reshape(8, 2, -1)[:, :, idx].numpy() - g[f"{tag}_center_samples"]).max()
    assert max(e1, e2) < 5e-5, (e1, e2)
    _, amax, arg = _oracle_chain(sdf, cen)
    report = []
    n = pc.check_peaks_against_fixture(g, tag, amax, arg, field_err=max(e1, e2, 1e-6), report=report)
    print(f"{tag}: {n} certified maps equal; field err {max(e1, e2):.2e}; uncertified differences: {report or 'none'}")


def test_oracle_forward_full_size_matches_reference(golden_dir):
    """benchmark-size forward (ViT-B/16 wiring, 384x384, B=1): 4096 sampled outputs + per-channel statistics"""
    g = np.load(os.path.join(golden_dir, "fwd_dpt_base_384_sampled.npz"))
    cfg = orc.CONFIGS["dpt_base"]
    sd = {k: torch.from_numpy(hash_init(k, s, "base")) for k, s in orc.state_dict_spec(cfg).items()}
    x = torch.from_numpy(synth.blob_images(1, 384, 384, seed=11))
    inter = {}
    with torch.no_grad():
        out = orc.forward(sd, x, cfg, inter=inter)
    idx = g["sample_idx"]
    cen, sdf, feat = out["center_fields"][0], out["sdf_maps"][0], inter["feat"][0]
    np.testing.assert_allclose(cen.reshape(2, -1)[:, idx].numpy(), g["center_samples"], atol=5e-5, rtol=0)
    np.testing.assert_allclose(sdf.reshape(1, -1)[:, idx].numpy(), g["sdf_samples"], atol=5e-5, rtol=0)
    np.testing.assert_allclose(cen.mean(dim=(1, 2)).numpy(), g["center_mean"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(sdf.abs().amax(dim=(1, 2)).numpy(), g["sdf_absmax"], atol=5e-5, rtol=0)
    np.testing.assert_allclose(feat.reshape(256, -1)[:, idx[:256]].numpy(), g["feat_samples"], atol=2e-4, rtol=1e-4)


def test_oracle_dpt_small_matches_reference_builders(golden_dir):
    """BASELINE configs[0] shape (ViT-S/16, 224x224, batch 2): the dpt_small extension assembled by the reference's own
    builders (_make_vit_b16_backbone / _make_scratch / _make_fusion_block) -- schema and sampled outputs"""
    g = np.load(os.path.join(golden_dir, "fwd_dpt_small_224_sampled.npz"))
    cfg = orc.CONFIGS["dpt_small"]
    spec = orc.state_dict_spec(cfg)
    ref_schema = [(l.split()[0], tuple(int(v) for v in l.split()[1:])) for l in open(os.path.join(golden_dir, "schema_dpt_small.txt"))]
    assert [(k, tuple(v)) for k, v in spec.items()] == ref_schema
    sd = {k: torch.from_numpy(hash_init(k, s, "dpt_small")) for k, s in spec.items()}
    x = torch.from_numpy(synth.blob_images(2, 224, 224, seed=12))
    with torch.no_grad():
        out = orc.forward(sd, x, cfg)
    idx = g["sample_idx"]
    np.testing.assert_allclose(out["center_fields"].reshape(2, 2, -1)[:, :, idx].numpy(), g["center_samples"], atol=5e-5, rtol=0)
    np.testing.assert_allclose(out["sdf_maps"].reshape(2, 1, -1)[:, :, idx].numpy(), g["sdf_samples"], atol=5e-5, rtol=0)
    np.testing.assert_allclose(out["center_fields"].mean(dim=(0, 2, 3)).numpy(), g["center_mean"], atol=2e-5, rtol=0)
