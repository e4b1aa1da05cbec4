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


def test_random_resized_crop_params_restatement():
    """RandomResizedCrop.get_params as restated in unmore_amd.labels (torchvision is absent: properties of the published
    algorithm, not a pinned stream): boxes fit the frame, area fraction and aspect ratio stay inside their ranges up to the
    integer rounding, the draw is reproducible from the generator, and the fallback is the ratio-clamped central crop."""
    import math
    import torch
    from unmore_amd.labels import random_resized_crop_params
    g = torch.Generator().manual_seed(0)
    boxes = [random_resized_crop_params(400, 400, scale=(0.08, 1.0), generator=g) for _ in range(300)]
    for (t, l, h, w) in boxes:
        assert 0 <= t <= 400 - h and 0 <= l <= 400 - w and 0 < h <= 400 and 0 < w <= 400
        frac = h * w / 160000.0
        assert 0.07 <= frac <= 1.01
        assert 0.73 <= w / h <= 1.37
    g2 = torch.Generator().manual_seed(0)
    assert boxes[:5] == [random_resized_crop_params(400, 400, scale=(0.08, 1.0), generator=g2) for _ in range(5)]
    assert len(set(boxes)) > 250
    # scale > 1 can never fit: ten failed tries, then the central crop of the whole (in-ratio) frame
    assert random_resized_crop_params(400, 400, scale=(4.0, 5.0), generator=g) == (0, 0, 400, 400)
    # a frame wider than the ratio range: full height, width = round(h * max ratio), centred
    t, l, h, w = random_resized_crop_params(100, 400, scale=(30.0, 40.0), generator=g)
    assert (h, w) == (100, int(round(100 * 1.33))) and t == 0 and l == (400 - w) // 2
    assert math.isclose(w / h, 1.33, rel_tol=0.01)


def test_oracle_crop_branch_reduces_to_plain_branch_for_the_full_frame():
    """With the whole 400x400 frame as the crop and image_size 400 every resize is the identity, so the random-crop branch
    (datasets.py:161-182) must give the labels of the plain branch (:183-190) for the 400x400 mask."""
    import numpy as np
    import torch
    from oracle import labels_oracle as LO
    rng = np.random.default_rng(2)
    yy, xx = np.mgrid[0:400, 0:400]
    mask = torch.from_numpy((((yy - 180) / 90.0) ** 2 + ((xx - 230) / 60.0) ** 2 <= 1).astype(np.float32))
    image = torch.from_numpy(rng.random((3, 400, 400)).astype(np.float32))
    img, lab = LO.training_item_random_crop(image, mask, (0, 0, 400, 400), 400)
    ref = LO.labels_from_mask(mask)
    assert torch.equal(img, image)
    assert torch.equal(lab["sdf"], ref["sdf"]) and torch.equal(lab["saliency_mask"], ref["saliency_mask"])
    torch.testing.assert_close(lab["center_field"], ref["center_field"], atol=1e-6, rtol=0)
