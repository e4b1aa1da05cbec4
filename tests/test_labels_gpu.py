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


# ---- the random-crop branch of the training item (datasets.py:144-190) ----------------------------------------------------
def _items(seed, sizes):
    rng = np.random.default_rng(seed)
    images, masks = [], []
    for (h, w) in sizes:
        images.append(torch.from_numpy(rng.random((3, h, w)).astype(np.float32)))
        yy, xx = np.mgrid[0:h, 0:w]
        cy, cx = rng.uniform(0.3, 0.7) * h, rng.uniform(0.3, 0.7) * w
        ry, rx = rng.uniform(h / 8 + 1, h / 3 + 1), rng.uniform(w / 8 + 1, w / 3 + 1)
        masks.append(torch.from_numpy((((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1).astype(np.float32)))
    return images, masks


def _check_items(images, masks, params, S, use_bg=True):
    from unmore_amd.labels import synthesize_training_items
    dev = _dev()
    img, lab, used = synthesize_training_items([i.to(dev) for i in images], [m.to(dev) for m in masks], S, use_bg_sdf=use_bg, params=params)
    assert used == params
    for b in range(len(images)):
        wi, wl = LO.training_item_random_crop(images[b], masks[b], params[b], S, use_bg)
        torch.testing.assert_close(img[b].cpu(), wi, atol=2e-6, rtol=0)                       # bilinear: same taps, f32 rounding order
        assert torch.equal(lab["instance_mask"][b].cpu(), wl["instance_mask"]), f"mask of item {b}"
        assert torch.equal(lab["saliency_mask"][b].cpu(), wl["saliency_mask"])
        torch.testing.assert_close(lab["object_center"][b].cpu(), wl["object_center"].float(), atol=1e-4, rtol=0)
        torch.testing.assert_close(lab["sdf"][b].cpu(), wl["sdf"], atol=2e-6, rtol=0)           # fg: resized (bilinear), bg: exact
        torch.testing.assert_close(lab["center_field"][b].cpu(), wl["center_field"], atol=2e-5, rtol=0)


def test_training_items_random_crop_branch():
    images, masks = _items(3, [(300, 420), (400, 400), (513, 257), (96, 128)])
    params = [(40, 60, 300, 280), (0, 0, 400, 400), (150, 120, 113, 150), (200, 10, 37, 49)]   # (top, left, h, w) in the 400x400 frame
    _check_items(images, masks, params, 128)
    _check_items(images, masks, params, 384, use_bg=False)


def test_training_items_object_cropped_out_and_empty_mask():
    images, masks = _items(4, [(200, 200), (240, 320), (180, 180)])
    masks[0] = torch.zeros_like(masks[0]); masks[0][20:60, 30:70] = 1     # object in the top-left corner ...
    masks[1] = torch.zeros_like(masks[1])                                  # ... empty mask: resized image + all-zero labels
    params = [(250, 250, 120, 120), (10, 10, 200, 200), (0, 0, 400, 400)]  # ... crop 0 misses the object entirely
    _check_items(images, masks, params, 96)


def test_training_items_draw_their_own_boxes():
    from unmore_amd.labels import synthesize_training_items
    dev = _dev()
    images, masks = _items(5, [(260, 300), (300, 260)])
    g = torch.Generator().manual_seed(11)
    img, lab, params = synthesize_training_items([i.to(dev) for i in images], [m.to(dev) for m in masks], 64, scale=(0.3, 1.0), generator=g)
    assert img.shape == (2, 3, 64, 64) and lab["sdf"].shape == (2, 64, 64)
    for (t, l, h, w) in params:
        assert 0 <= t and 0 <= l and t + h <= 400 and l + w <= 400 and h > 0 and w > 0
    _check_items(images, masks, params, 64)


def test_distance_transform_and_resizes_standalone():
    import torch.nn.functional as F
    from unmore_amd.labels import distance_transform, resize_bilinear, resize_nearest_u8
    dev = _dev()
    rng = np.random.default_rng(9)
    m = torch.from_numpy((rng.random((3, 57, 83)) > 0.3).astype(np.uint8))
    got = distance_transform(m.to(dev), normalize=False).cpu()
    for b in range(3):
        assert torch.equal(got[b], torch.from_numpy(LO.distance_transform_3x3(m[b].numpy())))
    x = torch.from_numpy(rng.random((2, 3, 57, 83)).astype(np.float32))
    torch.testing.assert_close(resize_bilinear(x.to(dev), 40, 101).cpu(), F.interpolate(x, size=(40, 101), mode="bilinear", align_corners=False),
                               atol=2e-6, rtol=0)
    for size in ((40, 101), (400, 400), (19, 7)):
        assert torch.equal(resize_nearest_u8(m.to(dev), *size).cpu(), F.interpolate(m.unsqueeze(1).float(), size=size, mode="nearest")[:, 0].to(torch.uint8))
