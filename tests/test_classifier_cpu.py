"""Existence classifier: oracle vs the fixtures made by the reference's own `Binary_Classifier`
class, and the drop-in module's checkpoint schema.  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import classifier_oracle as CO
from unmore_amd.hashrng import uniform, uniform01

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _schema():
    out = []
    for line in open(os.path.join(GOLD, "clf_schema.txt")):
        parts = line.split()
        out.append((parts[0], tuple(int(v) for v in parts[1:])))
    return out


def test_oracle_schema_matches_reference_object():
    assert CO.state_dict_spec() == _schema()
    assert len(_schema()) == 322


@pytest.mark.parametrize("B,S", [(2, 64), (3, 128)])
def test_oracle_forward_matches_reference_fixture(B, S):
    sd = CO.hash_state("clf", uniform)
    x = torch.from_numpy(uniform01(f"img:clf{S}", (B, 3, S, S)))
    want = np.load(os.path.join(GOLD, f"clf_fwd_{S}.npz"))["prob"]
    got = CO.forward(sd, x).numpy()
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-6)


def test_dropin_module_schema_and_guards():
    from unmore_amd.binary_classifier import Binary_Classifier
    net = Binary_Classifier(device="cpu", image_size=128, args=None)
    sd = net.state_dict()
    assert [(k, tuple(v.shape)) for k, v in sd.items()] == _schema()
    net.load_state_dict(CO.hash_state("clf", uniform), strict=True)   # a reference-format checkpoint loads strictly
    assert net.image_size == 128 and net.device == "cpu" and hasattr(net, "classifier_backbone") and hasattr(net, "binary_classification_head")
    net.eval()
    with torch.no_grad(), pytest.raises(RuntimeError):
        net(torch.zeros(1, 3, 64, 64))   # no CPU fallback
