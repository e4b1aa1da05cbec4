"""Ground-truth synthesis on the MI355X against the CPU oracle: bit-exact distance fields / saliency,
centre field within 1e-6."""
import numpy as np
import pytest
import torch

from oracle import labels_oracle as LO

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU test run without a GPU"
    return torch.device("cuda:0")


def _check(masks, centers=None, use_bg=True, dt=LO.distance_transform_3x3):
    from unmore_amd.labels import synthesize_labels
    got = synthesize_labels(masks.to(_dev()), None if centers is None else centers.to(_dev()), use_bg)
    for b in range(masks.shape[0]):
        want = LO.labels_from_mask(masks[b], None if centers is None else centers[b], use_bg, dt)
        assert torch.equal(got["sdf"][b].cpu(), want["sdf"]), f"sdf of image {b}"
        assert torch.equal(got["saliency_mask"][b].cpu(), want["saliency_mask"])
        torch.testing.assert_close(got["center_field"][b].cpu(), want["center_field"], atol=1e-6, rtol=0)


def test_small_masks_vs_literal_raster():
    rng = np.random.default_rng(1)
    masks = torch.from_numpy((rng.random((6, 24, 31)) > 0.4).astype(np.uint8))
    masks[4] = 0            # empty mask: all-zero labels (datasets.py:128-138)
    masks[5] = 1            # no background pixel at all
    _check(masks, dt=LO.distance_transform_3x3_literal)
    _check(masks, use_bg=False, dt=LO.distance_transform_3x3_literal)


@pytest.mark.parametrize("H,W", [(384, 384), (97, 513), (33, 1)])
def test_full_size_ellipses_and_ragged(H, W):
    rng = np.random.default_rng(H + W)
    B = 3
    yy, xx = np.mgrid[0:H, 0:W]
    masks = []
    for b in range(B):
        cy, cx = rng.uniform(0.25, 0.75) * H, rng.uniform(0.25, 0.75) * W
        ry, rx = rng.uniform(H / 8 + 1, H / 3 + 1), rng.uniform(W / 8 + 1, W / 3 + 1)
        masks.append((((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1).astype(np.uint8))
    masks = torch.from_numpy(np.stack(masks))
    _check(masks)
    centers = torch.tensor([[W * 0.4, H * 0.55]] * B, dtype=torch.float32)   # explicit (x, y) centres (datasets.py:171-173)
    _check(masks, centers)
