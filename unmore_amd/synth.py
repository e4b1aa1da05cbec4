"""Synthetic training batches of the shape the benchmark needs:
uniform-[0,1) RGB images and, per image, one filled ellipse pseudo-mask from which the
labels are derived with the formulas of the reference's dataset code (datasets.py:158-222):
saliency = mask; center field = unit vector from the mask's bbox centre (channel 0 = row
offset) inside the mask, 0 outside; boundary distance = DT(fg)/max - DT(bg)/max (--use_bg_sdf).
Host-side numpy/scipy, run once before timing (the data loader is out of scope)."""
import numpy as np
from scipy import ndimage

from .hashrng import uniform01


def ellipse_masks(B, H, W, seed=0):
    u = uniform01(f"ellipse:{seed}", (B, 4))
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    masks = np.zeros((B, H, W), np.uint8)
    for b in range(B):
        cy, cx = (0.25 + 0.5 * u[b, 0]) * H, (0.25 + 0.5 * u[b, 1]) * W
        ry, rx = H / 8 + u[b, 2] * (H / 3 - H / 8), W / 8 + u[b, 3] * (W / 3 - W / 8)
        masks[b] = (((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2) <= 1.0
    return masks


def labels_from_masks(masks):
    B, H, W = masks.shape
    cf = np.zeros((B, 2, H, W), np.float32)
    sdf = np.zeros((B, 1, H, W), np.float32)
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    for b in range(B):
        m = masks[b] > 0
        ys, xs = np.nonzero(m)
        cy, cx = (ys.min() + ys.max()) / 2.0, (xs.min() + xs.max()) / 2.0
        oy, ox = yy - cy, xx - cx
        n = np.sqrt(oy * oy + ox * ox) + 1e-12
        cf[b, 0] = np.where(m, oy / n, 0)
        cf[b, 1] = np.where(m, ox / n, 0)
        fg = ndimage.distance_transform_edt(m).astype(np.float32)
        bg = ndimage.distance_transform_edt(~m).astype(np.float32)
        sdf[b, 0] = fg / max(float(fg.max()), 1e-6) - bg / max(float(bg.max()), 1e-6)
    return cf, sdf, masks.astype(np.float32)[:, None]


def make_batch(B, H, W, seed=0):
    """(images [B,3,H,W], center_field [B,2,H,W], sdf [B,1,H,W], saliency [B,1,H,W]) float32 numpy."""
    images = uniform01(f"images:{seed}", (B, 3, H, W))
    cf, sdf, sal = labels_from_masks(ellipse_masks(B, H, W, seed))
    return images, cf, sdf, sal
