"""Ground-truth synthesis on the device: the label half of the reference's dataset item
(datasets.py:158-159,171-222, the branch without random crop) for a batch of masks, one call instead of a cv2 / PyTorch
pipeline per image on the host."""
import torch

from . import _lib as L
from .ops import _p, _stream, _need_gpu


def synthesize_labels(masks, object_centers=None, use_bg_sdf=True):
    """masks: [B,H,W] bool / integer / float tensor on the GPU (non-zero = object), already at the training resolution.
    object_centers: optional [B,2] (x, y) float tensor in the same pixel coordinates (the reference scales the centre of
    the pre-resize mask, datasets.py:171-173); None = bounding-box centre of each mask (datasets.py:158-159).
    Returns the reference's label dict (datasets.py:210-216) batched: 'center_field' [B,2,H,W] f32, 'saliency_mask'
    [B,H,W] f32 (0/1; the collate casts everything to float, datasets.py:70-75), 'sdf' [B,H,W] f32."""
    _need_gpu(masks)
    assert masks.dim() == 3
    B, H, W = masks.shape
    m8 = (masks != 0).to(torch.uint8).contiguous()
    dev = masks.device
    cen = None
    if object_centers is not None:
        cen = object_centers.to(dev, torch.float32).contiguous()
        assert cen.shape == (B, 2)
    cf = torch.empty((B, 2, H, W), dtype=torch.float32, device=dev)
    sal = torch.empty((B, H, W), dtype=torch.float32, device=dev)
    sdf = torch.empty((B, H, W), dtype=torch.float32, device=dev)
    ws_bytes = L.lib().umr_label_synthesis_workspace(B, H, W)
    ws = torch.empty((ws_bytes + 3) // 4, dtype=torch.int32, device=dev)
    L.check(L.lib().umr_label_synthesis(_p(m8), _p(cen), _p(cf), _p(sal), _p(sdf), _p(ws), ws_bytes, B, H, W, int(bool(use_bg_sdf)),
                                        _stream()), "umr_label_synthesis")
    return {"center_field": cf, "saliency_mask": sal, "sdf": sdf}


# ---------------------------------------------------------------------------------------------------------------------
# The random-crop branch of the training item (datasets.py:144-190; __getitem__ always takes it, :109)
# ---------------------------------------------------------------------------------------------------------------------
import math

PRE_CROP_SIZE = 400  # datasets.py:103-104


def random_resized_crop_params(height, width, scale=(0.08, 1.0), ratio=(0.75, 1.33), generator=None):
    """torchvision.transforms.RandomResizedCrop.get_params (torchvision 0.14.1, the version README.md:25 names; called at
    datasets.py:167-168) restated: ten tries of (area fraction ~ U(scale), log aspect ~ U(log ratio)), then a position
    uniform over the placements that fit; falls back to the central crop closest to the ratio range.  Returns
    (top, left, h, w).  The torchvision package is absent from this image: the algorithm follows its published source, the
    random stream is torch's CPU generator as there (parity at this boundary is unpinned)."""
    area = height * width
    log_lo, log_hi = math.log(ratio[0]), math.log(ratio[1])
    for _ in range(10):
        target_area = area * torch.empty(1).uniform_(scale[0], scale[1], generator=generator).item()
        aspect = math.exp(torch.empty(1).uniform_(log_lo, log_hi, generator=generator).item())
        w = int(round(math.sqrt(target_area * aspect)))
        h = int(round(math.sqrt(target_area / aspect)))
        if 0 < w <= width and 0 < h <= height:
            i = int(torch.randint(0, height - h + 1, size=(1,), generator=generator).item())
            j = int(torch.randint(0, width - w + 1, size=(1,), generator=generator).item())
            return i, j, h, w
    in_ratio = float(width) / float(height)
    if in_ratio < min(ratio):
        w = width
        h = int(round(w / min(ratio)))
    elif in_ratio > max(ratio):
        h = height
        w = int(round(h * max(ratio)))
    else:
        w, h = width, height
    return (height - h) // 2, (width - w) // 2, h, w


def _crop_resize_batch(src, boxes, Ho, Wo, nearest):
    B, C, H, W = src.shape
    dst = torch.empty((B, C, Ho, Wo), dtype=src.dtype, device=src.device)
    bx = torch.as_tensor(boxes, dtype=torch.int32).reshape(B, 4).to(src.device)
    L.check(L.lib().umr_crop_resize_batch(_p(src), _p(bx), _p(dst), B, C, H, W, Ho, Wo, int(nearest), _stream()), "umr_crop_resize_batch")
    return dst


def resize_bilinear(x, Ho, Wo):
    """transforms.Resize((Ho, Wo), BILINEAR) on a float tensor [B,C,H,W] (no antialias, datasets.py:99,103)."""
    _need_gpu(x)
    x = x.contiguous().float()
    B, _, H, W = x.shape
    return _crop_resize_batch(x, [[0, 0, W, H]] * B, Ho, Wo, False)


def resize_nearest_u8(m, Ho, Wo):
    """transforms.Resize((Ho, Wo), NEAREST) on a mask [B,H,W] (any dtype; returned as u8 0/1, datasets.py:100,104)."""
    _need_gpu(m)
    m8 = (m != 0).to(torch.uint8).contiguous()
    B, H, W = m8.shape
    return _crop_resize_batch(m8.view(B, 1, H, W), [[0, 0, W, H]] * B, Ho, Wo, True).view(B, Ho, Wo)


def distance_transform(masks, normalize=True):
    """cv2.distanceTransform(u8, DIST_L2, 3) per mask [B,H,W] (distance to the nearest zero pixel), optionally divided by its
    maximum (datasets.py:162-163)."""
    _need_gpu(masks)
    m8 = (masks != 0).to(torch.uint8).contiguous()
    B, H, W = m8.shape
    out = torch.empty((B, H, W), dtype=torch.float32, device=m8.device)
    nbytes = L.lib().umr_distance_transform_workspace(B, H, W)
    ws = torch.empty((nbytes + 3) // 4, dtype=torch.int32, device=m8.device)
    L.check(L.lib().umr_distance_transform(_p(m8), _p(out), _p(ws), nbytes, B, H, W, int(bool(normalize)), _stream()), "umr_distance_transform")
    return out


def synthesize_training_items(images, masks, image_size, scale=(0.08, 1.0), use_bg_sdf=True, generator=None, params=None):
    """A batch of the reference's training items (datasets.py:111-216 after file decoding, random_crop=True as __getitem__
    uses it) on the device.
    images: list of [3,h,w] float tensors in [0,1] on the GPU (sizes may differ); masks: list of [h,w] tensors (non-zero =
    object).  scale = (args.random_crop_scale_min, args.random_crop_scale_max) (train_objectness_net.py:813-814).
    params: optional list of (top, left, h, w) crop boxes in the 400x400 frame (tests); None = draw them with
    `random_resized_crop_params` from `generator` (torch CPU generator; None = the global one), one item after the other as a
    single-worker DataLoader would.
    Returns (images [B,3,S,S] f32, labels dict: 'center_field' [B,2,S,S], 'saliency_mask' [B,S,S], 'instance_mask' [B,S,S],
    'object_center' [B,2] (x, y), 'sdf' [B,S,S], all f32 as after the reference's collate) and the crop boxes used.
    Items whose 400x400 mask is empty get the resized image and all-zero labels (datasets.py:146-157)."""
    B, S, P = len(images), int(image_size), PRE_CROP_SIZE
    assert B == len(masks) and B > 0
    dev = images[0].device
    img400 = torch.cat([resize_bilinear(im.unsqueeze(0), P, P) for im in images])                 # datasets.py:144
    m400 = torch.cat([resize_nearest_u8(mk.unsqueeze(0), P, P) for mk in masks])                    # :145
    # bounding-box centre of the 400x400 mask (:158-159) and emptiness (:146-147) -- small reductions, done with torch
    mb = m400.bool()
    rows, cols = mb.any(dim=2), mb.any(dim=1)
    empty = ~rows.any(dim=1)
    ar = torch.arange(P, device=dev)
    big = P + 1
    y0 = torch.where(rows, ar, big).amin(dim=1).float(); y1 = torch.where(rows, ar, -1).amax(dim=1).float()
    x0 = torch.where(cols, ar, big).amin(dim=1).float(); x1 = torch.where(cols, ar, -1).amax(dim=1).float()
    cx, cy = (x0 + x1) / 2, (y0 + y1) / 2
    fg400 = distance_transform(m400, normalize=True)                                                # :162-164
    if params is None:
        params = [random_resized_crop_params(P, P, scale=scale, generator=generator) for _ in range(B)]   # :167-168
    assert len(params) == B
    pt = torch.tensor(params, dtype=torch.float32, device=dev)                                      # top, left, h, w
    boxes = [[l, t, l + w, t + h] for (t, l, h, w) in params]
    alld = torch.cat([img400, fg400.unsqueeze(1)], dim=1)                                           # image + sdf share the bilinear resize (:174,176)
    out = _crop_resize_batch(alld, boxes, S, S, False)
    m_s = _crop_resize_batch(m400.view(B, 1, P, P), boxes, S, S, True).view(B, S, S)                # :175
    img_s, fg_s = out[:, :3].contiguous(), out[:, 3].contiguous()
    # :180-182 -- (centre - offset) * (image_size / extent): the factor is a Python float there, applied as a float32 scalar
    fx = torch.tensor([S / w for (_, _, _, w) in params], dtype=torch.float64).to(torch.float32).to(dev)
    fy = torch.tensor([S / h for (_, _, h, _) in params], dtype=torch.float64).to(torch.float32).to(dev)
    ocx = (cx - pt[:, 1]) * fx
    ocy = (cy - pt[:, 0]) * fy
    centers = torch.stack([ocx, ocy], dim=1).contiguous()
    cf = torch.empty((B, 2, S, S), dtype=torch.float32, device=dev)
    sal = torch.empty((B, S, S), dtype=torch.float32, device=dev)
    sdf = torch.empty((B, S, S), dtype=torch.float32, device=dev)
    nbytes = L.lib().umr_label_synthesis_workspace(B, S, S)
    ws = torch.empty((nbytes + 3) // 4, dtype=torch.int32, device=dev)
    L.check(L.lib().umr_label_synthesis_cropped(_p(m_s), _p(centers), _p(fg_s), _p(cf), _p(sal), _p(sdf), _p(ws), nbytes, B, S, S,
                                                int(bool(use_bg_sdf)), _stream()), "umr_label_synthesis_cropped")
    inst = m_s.float()
    if bool(empty.any()):   # datasets.py:146-157: image resized directly, all-zero labels
        e = empty.nonzero().flatten()
        img_s[e] = resize_bilinear(img400[e], S, S)
        cf[e] = 0; sal[e] = 0; sdf[e] = 0; inst[e] = 0; centers[e] = 0
    labels = {"center_field": cf, "saliency_mask": sal, "instance_mask": inst, "object_center": centers, "sdf": sdf}
    return img_s, labels, params
