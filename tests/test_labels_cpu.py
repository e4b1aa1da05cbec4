"""Ground-truth synthesis oracle: the scan form of the 3x3 chamfer transform equals the literal
two-pass raster form (OpenCV's distanceTransform_3x3 as restated; parity UNPINNED -- cv2 is absent, see the oracle header)."""
import numpy as np
import pytest
import torch

from oracle import labels_oracle as LO


@pytest.mark.parametrize("shape,p", [((9, 13), 0.35), ((24, 31), 0.1), ((1, 7), 0.5), ((6, 1), 0.5), ((17, 17), 0.9)])
def test_scan_form_equals_literal_raster(shape, p):
    rng = np.random.default_rng(sum(shape))
    m = (rng.random(shape) > p).astype(np.uint8)
    assert np.array_equal(LO.distance_transform_3x3_literal(m), LO.distance_transform_3x3(m))


def test_known_values_and_label_semantics():
    m = np.zeros((7, 9), np.uint8)
    m[2:6, 3:8] = 1
    d = LO.distance_transform_3x3_literal(m)
    assert d[2, 3] == np.float32(62587 / 65536) and d[3, 4] == np.float32(2 * 62587 / 65536) and d[0, 0] == 0
    lab = LO.labels_from_mask(torch.from_numpy(m))
    assert lab["sdf"].max() == 1.0 and lab["sdf"].min() == -1.0
    cf = lab["center_field"]
    assert torch.all(cf[:, m == 0] == 0)
    nrm = torch.linalg.norm(cf[:, 2:6, 3:8], dim=0)
    assert torch.allclose(nrm[nrm > 0], torch.ones_like(nrm[nrm > 0]), atol=1e-6)
    # bbox centre (x, y) = ((3+7)/2, (2+5)/2) = (5, 3.5): the pixel row 2 / col 5 points straight up (negative row offset)
    assert cf[0, 2, 5] < 0 and abs(float(cf[1, 2, 5])) < 1e-6
    empty = LO.labels_from_mask(torch.zeros(5, 6))
    assert all(float(v.abs().max()) == 0.0 for v in empty.values())
