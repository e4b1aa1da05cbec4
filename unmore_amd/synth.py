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


def blob_images(B, H, W, seed=0, n_blobs=3):
    """Structured RGB test images: a dim flat background and `n_blobs` flat-coloured discs per image (object-like content, so
    that a randomly initialised net's fields vary over the image instead of averaging out as they do on uniform noise).
    Determined by (B, H, W, seed) through the hash RNG alone."""
    u = uniform01(f"blobs:{seed}", (B, 3 + 6 * n_blobs))
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    img = np.zeros((B, 3, H, W), np.float32)
    for b in range(B):
        img[b] = (np.float32(0.2) * u[b, 0:3])[:, None, None]
        for k in range(n_blobs):
            q = u[b, 3 + 6 * k: 9 + 6 * k]
            cy, cx = (np.float32(0.2) + np.float32(0.6) * q[0]) * H, (np.float32(0.2) + np.float32(0.6) * q[1]) * W
            r = (np.float32(0.1) + np.float32(0.2) * q[2]) * min(H, W)
            inside = ((yy - cy) ** 2 + (xx - cx) ** 2) < r * r
            img[b] = np.where(inside[None], q[3:6][:, None, None], img[b])
    return img


def object_like_fields(B, H, W, seed=0):
    """Synthetic (boundary-distance [B,H,W], centre-field [B,2,H,W]) maps with the structure the reasoning stage sees: two
    discs per map with a soft-sign distance profile (only correctly rounded IEEE operations, so the maps regenerate bit-exactly
    on any host) and unit vectors converging on (even maps) or diverging from (odd maps) the
    disc centres, plus small hash noise.  Inputs for the peak-picking fixtures (tests/golden/make_golden_r2.py)."""
    u = uniform01(f"fields:{seed}", (B, 2, 3))
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    sdf = np.zeros((B, H, W), np.float32)
    cen = np.zeros((B, 2, H, W), np.float32)
    for b in range(B):
        sign = np.float32(-1.0 if b % 2 == 0 else 1.0)
        for k in range(2):
            cy, cx = (np.float32(0.25) + np.float32(0.5) * u[b, k, 0]) * H, (np.float32(0.25) + np.float32(0.5) * u[b, k, 1]) * W
            r = (np.float32(0.15) + np.float32(0.2) * u[b, k, 2]) * min(H, W)
            d = np.sqrt((yy - cy) ** 2 + (xx - cx) ** 2).astype(np.float32)
            t = (r - d) / np.float32(8.0)
            sdf[b] = np.maximum(sdf[b], (t / (np.float32(1.0) + np.abs(t))).astype(np.float32))  # soft sign: +,-,*,/,sqrt only
            inside = d < r
            n = d + np.float32(1e-6)
            cen[b, 0] = np.where(inside, sign * (yy - cy) / n, cen[b, 0])
            cen[b, 1] = np.where(inside, sign * (xx - cx) / n, cen[b, 1])
        sdf[b] = sdf[b] * np.float32(2.0) - np.float32(0.3)
    sdf += np.float32(0.1) * (uniform01(f"fields:{seed}:ns", (B, H, W)) - np.float32(0.5))
    cen += np.float32(0.1) * (uniform01(f"fields:{seed}:nc", (B, 2, H, W)) - np.float32(0.5))
    return sdf.astype(np.float32), cen.astype(np.float32)


# A randomly initialised net on noise gives identically empty eroded masks (every score map is zero and peak picking has
# nothing to do).  The committed peak fixtures (tests/golden/make_golden_r2.py, EDITS) therefore use two documented weight
# edits per hash-weight set that give surviving masks of varied size on the blob images: the last boundary-distance bias +=
# shift (pre-tanh), the last centre-field layer *= scale.  bench.py --workload cfg5 uses the same nets.
PEAK_EDITS = {"base": (0.5, 1.5), "tiny": (0.05, 2.0)}


def peak_edited_state_dict(spec, wtag):
    """hash weights `wtag` + PEAK_EDITS[wtag]; spec: name -> shape.  numpy f32 arrays."""
    from .hashrng import hash_init
    shift, scale = PEAK_EDITS[wtag]
    sd = {k: hash_init(k, tuple(s), wtag) for k, s in spec.items()}
    sd["sdf_prediction_head.3.bias"] = sd["sdf_prediction_head.3.bias"] + np.float32(shift)
    sd["center_field_prediction_head.6.weight"] = sd["center_field_prediction_head.6.weight"] * np.float32(scale)
    sd["center_field_prediction_head.6.bias"] = sd["center_field_prediction_head.6.bias"] * np.float32(scale)
    return sd


def reasoning_scene(H, W, seed=0, n_objects=4):
    """A [3,H,W] float32 "image" whose channels ARE object-like fields, for exercising the object-reasoning loop
    (object_reasoning.py:615-665) end to end without a trained net (tests/discovery_stubs.py reads them back out of the resized
    crops): channel 0 = a boundary-distance-like field in (-1, 1), positive inside `n_objects` ellipses and falling off outside;
    channels 1, 2 = the unit vector (row, column offset) from the nearest ellipse's centre inside the ellipses, zero outside.
    Deterministic numpy (hashrng), so the golden generator and the tests rebuild the same array."""
    u = uniform01(f"scene:{seed}", (n_objects, 4))
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    field = np.full((H, W), -1.0, np.float32)
    cf = np.zeros((2, H, W), np.float32)
    best = np.full((H, W), np.inf, np.float32)
    for k in range(n_objects):
        cy, cx = np.float32((0.15 + 0.7 * u[k, 0]) * H), np.float32((0.15 + 0.7 * u[k, 1]) * W)
        ry, rx = np.float32(H / 14 + u[k, 2] * H / 7), np.float32(W / 14 + u[k, 3] * W / 7)
        oy, ox = yy - cy, xx - cx
        r = np.sqrt((oy / ry) ** 2 + (ox / rx) ** 2).astype(np.float32)          # 1 on the ellipse
        f = np.tanh(np.float32(2.5) * (np.float32(1.0) - r)).astype(np.float32)
        field = np.maximum(field, f)
        n = (np.sqrt(oy * oy + ox * ox) + np.float32(1e-6)).astype(np.float32)
        near = r < best
        inside = r <= 1.0
        cf[0] = np.where(near & inside, oy / n, np.where(near, 0.0, cf[0]))
        cf[1] = np.where(near & inside, ox / n, np.where(near, 0.0, cf[1]))
        best = np.minimum(best, r)
    return np.concatenate([field[None], cf], axis=0).astype(np.float32)
