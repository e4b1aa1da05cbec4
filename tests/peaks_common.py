"""Shared pieces of the peak-index parity tests (fixture class (iv) of SURVEY.md section 8c, tests/golden/peaks.npz made by
tests/golden/make_golden_r2.py from the reference's own functions): the documented weight edits that give a randomly
initialised net surviving masks, and the comparison rules."""
import os

import numpy as np
import torch

from unmore_amd import synth
from unmore_amd.hashrng import hash_init

HERE = os.path.dirname(os.path.abspath(__file__))
E2E = {"e2e_base128": ("dpt_base", "base"), "e2e_tiny128": ("dpt_tiny", "tiny")}
SYN = {"syn128": (6, 128, 128, 0), "syn96x160": (4, 96, 160, 1)}
SQRT2 = float(np.sqrt(2.0))


def load():
    return np.load(os.path.join(HERE, "golden", "peaks.npz"))


def edited_state_dict(spec, wtag, shift, scale):
    """hash weights + the fixture's edits (make_golden_r2.EDITS): last sdf bias += shift, last centre layer *= scale"""
    sd = {k: torch.from_numpy(hash_init(k, tuple(s), wtag)) for k, s in spec.items()}
    sd["sdf_prediction_head.3.bias"] = sd["sdf_prediction_head.3.bias"] + np.float32(shift)
    sd["center_field_prediction_head.6.weight"] = sd["center_field_prediction_head.6.weight"] * np.float32(scale)
    sd["center_field_prediction_head.6.bias"] = sd["center_field_prediction_head.6.bias"] * np.float32(scale)
    return sd


def e2e_images(tag):
    return torch.from_numpy(synth.blob_images(8, 128, 128, seed=7))


def eroded_mask(g, tag, B, HW):
    return np.unpackbits(g[f"{tag}_eroded_bits"], axis=1)[:, :HW].astype(bool)


def check_peaks_against_fixture(g, tag, amax, argmax, field_err, report):
    """amax [B] f64, argmax [B] int64 from the path under test whose fields are within `field_err` of the reference's.
    Non-vacuity: the fixture's maps with a peak must have a peak here.  Equality is REQUIRED wherever the fixture certifies the
    argmax for perturbations up to meta_cert_eps (>= field_err); elsewhere a difference is reported, not hidden."""
    ref_amax, ref_arg, cert = g[f"{tag}_amax"], g[f"{tag}_argmax"], g[f"{tag}_argmax_certified"]
    eps = float(g["meta_cert_eps"])
    assert field_err < eps, f"field error {field_err} is not below the certificate's epsilon {eps}"
    assert (ref_amax > 0).sum() >= len(ref_amax) // 2, "fixture is (nearly) vacuous"
    n_cert = 0
    for b in range(len(ref_amax)):
        same = int(argmax[b]) == int(ref_arg[b])
        if cert[b]:
            n_cert += 1
            assert ref_amax[b] > 0 and amax[b] > 0, f"{tag} map {b}: no peak"
            assert same, f"{tag} map {b}: argmax {int(argmax[b])} vs reference {int(ref_arg[b])} although certified"
            assert abs(float(amax[b]) - float(ref_amax[b])) <= SQRT2 * field_err + 1e-12
        elif not same:
            report.append(f"{tag} map {b}: argmax {int(argmax[b])} vs reference {int(ref_arg[b])} (uncertified: top-2 margin "
                          f"{g[f'{tag}_top2_margin'][b]:.3e}, {int(g[f'{tag}_flippable_pixels'][b])} flippable mask pixels)")
    assert n_cert >= 3, "too few certified maps"
    return n_cert
