"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY (ground-truth synthesis).

Restates the label half of the reference's dataset item (datasets.py:158-159,171-222, branch without random crop).  Only
`tests/` may import this file; the product (`unmore_amd/`) never does.

Parity status: UNPINNED.  The reference computes the distance fields with `cv2.distanceTransform(u8, cv2.DIST_L2, 3)`
(datasets.py:176,186); OpenCV (opencv-python==4.10.0.84, requirements.txt) is absent from /root/reference and from this
image, the dataset class cannot be imported without it, and the reference holds no fixtures.  `distance_transform_3x3`
restates OpenCV's published algorithm for that call (imgproc/src/distransform.cpp `distanceTransform_3x3`: two-pass 3x3
chamfer in 16.16 fixed point with the documented DIST_L2 3x3 weights a = 0.955, b = 1.3693, a one-pixel border of
INT_MAX >> 2, output clamped to INT_MAX >> 2 and scaled by 2^-16).  The centre-field lines are the reference's own PyTorch
expressions restated one to one.  `training_item_random_crop` restates the random-crop branch (datasets.py:144-190) for a
GIVEN crop box: the resizes are torchvision 0.14.1's tensor path (F.interpolate, no antialias), the box draw
(RandomResizedCrop.get_params) lies outside the restatement -- torchvision is absent too, that boundary is unpinned as well.
"""
import numpy as np
import torch
import torch.nn.functional as F

HV = int(round(0.955 * 65536))     # 62587
DG = int(round(1.3693 * 65536))    # 89738
INIT0 = (2 ** 31 - 1) >> 2


def distance_transform_3x3_literal(src):
    """The two raster passes exactly as written in OpenCV (pure Python loops: small inputs only)."""
    H, W = src.shape
    tmp = [[INIT0] * (W + 2) for _ in range(H + 2)]
    for i in range(H):
        row, up = tmp[i + 1], tmp[i]
        for j in range(W):
            if not src[i, j]:
                row[j + 1] = 0
            else:
                row[j + 1] = min(up[j] + DG, up[j + 1] + HV, up[j + 2] + DG, row[j] + HV)
    out = np.zeros((H, W), np.float32)
    scale = np.float32(1.0 / 65536.0)
    for i in range(H - 1, -1, -1):
        row, dn = tmp[i + 1], tmp[i + 2]
        for j in range(W - 1, -1, -1):
            t0 = row[j + 1]
            if t0 > HV:
                t0 = min(t0, dn[j + 2] + DG, dn[j + 1] + HV, dn[j] + DG, row[j + 2] + HV)
                row[j + 1] = t0
            out[i, j] = np.float32(min(t0, INIT0)) * scale
    return out


def distance_transform_3x3(src):
    """Same result with the in-row recurrences as min-plus scans (numpy, int64): any size."""
    src = np.asarray(src) != 0
    H, W = src.shape
    jj = np.arange(W, dtype=np.int64) * HV
    tmp = np.empty((H, W), np.int64)
    prev = np.full(W + 2, INIT0, np.int64)
    for i in range(H):
        c = np.minimum(np.minimum(prev[:-2] + DG, prev[1:-1] + HV), prev[2:] + DG)
        c[0] = min(c[0], INIT0 + HV)
        c = np.where(src[i], c, 0)
        row = np.minimum.accumulate(c - jj) + jj
        tmp[i] = row
        prev[1:-1] = row
    prev[:] = INIT0
    out = np.empty((H, W), np.float32)
    for i in range(H - 1, -1, -1):
        c = np.minimum(tmp[i], np.minimum(np.minimum(prev[2:] + DG, prev[1:-1] + HV), prev[:-2] + DG))
        c[-1] = min(c[-1], INIT0 + HV)
        row = np.minimum.accumulate((c + jj)[::-1])[::-1] - jj
        prev[1:-1] = row
        out[i] = np.minimum(row, INIT0).astype(np.float32) * np.float32(1.0 / 65536.0)
    return out


def labels_from_mask(mask, object_center=None, use_bg_sdf=True, dt=distance_transform_3x3):
    """datasets.py:158-159,176-216 for one mask [H,W] (torch, 0/1).  object_center = (x, y) or None (bbox centre)."""
    mask = (mask != 0).to(torch.int)
    H, W = mask.shape
    y, x = torch.where(mask > 0)
    if len(y) == 0:  # datasets.py:128-138
        return {"center_field": torch.zeros(2, H, W), "saliency_mask": torch.zeros(H, W), "sdf": torch.zeros(H, W)}
    if object_center is None:
        object_center = torch.tensor([(torch.min(x) + torch.max(x)) / 2, (torch.min(y) + torch.max(y)) / 2])
    sdf = dt(np.uint8(mask.numpy()))
    sdf = sdf / sdf.max() if sdf.max() > 0 else sdf
    sdf = torch.tensor(sdf)
    if use_bg_sdf:
        bg_mask = torch.where(mask == 0, 1, 0)
        bg_sdf = dt(np.uint8(bg_mask.numpy()))
        bg_sdf = bg_sdf / bg_sdf.max() if bg_sdf.max() > 0 else bg_sdf
        sdf = sdf + torch.tensor(bg_sdf) * (-1)
    xv, yv = torch.meshgrid([torch.arange(H), torch.arange(W)], indexing="ij")
    grid = torch.stack((xv, yv), 2).float().permute(2, 0, 1)
    ocf = grid - torch.tensor([float(object_center[1]), float(object_center[0])]).unsqueeze(1).unsqueeze(1)
    ocf = F.normalize(ocf, dim=0)
    center_field = torch.zeros_like(grid) + torch.where(mask > 0, 1, 0) * ocf
    center_field = F.normalize(center_field, dim=0)
    return {"center_field": center_field, "saliency_mask": torch.where(mask > 0, 1, 0).float(), "sdf": sdf}


# ---- the random-crop branch (datasets.py:144-190; __getitem__ uses it, :109) -------------------------------------------
def resize(x, size, nearest=False):
    """transforms.Resize(size, BILINEAR / NEAREST) on a tensor [C,H,W]: torchvision 0.14.1 resizes tensors with
    F.interpolate (no antialias by default; align_corners=False for bilinear)."""
    x4 = x.unsqueeze(0).float()
    if nearest:
        return F.interpolate(x4, size=size, mode="nearest")[0]
    return F.interpolate(x4, size=size, mode="bilinear", align_corners=False)[0]


def training_item_random_crop(image, mask, params, image_size, use_bg_sdf=True, dt=distance_transform_3x3):
    """datasets.py:144-216 for one decoded item with random_crop=True and GIVEN crop box params = (top, left, h, w) in the
    400x400 frame (the draw itself, torchvision's RandomResizedCrop.get_params, is outside this restatement).
    image [3,h,w] float in [0,1], mask [h,w] 0/1.  Returns (image [3,S,S], labels dict)."""
    S = image_size
    image = resize(image, (400, 400))                                                     # :144
    mask = resize(mask.unsqueeze(0), (400, 400), nearest=True)[0]                         # :145
    y, x = torch.where(mask > 0)
    if len(y) == 0 or len(x) == 0:                                                         # :146-157
        return resize(image, (S, S)), {"center_field": torch.zeros(2, S, S), "saliency_mask": torch.zeros(S, S),
                                       "instance_mask": torch.zeros(S, S), "object_center": torch.tensor([0.0, 0.0]),
                                       "sdf": torch.zeros(S, S)}
    obj_x_center = (torch.min(x) + torch.max(x)) / 2                                       # :158-159
    obj_y_center = (torch.min(y) + torch.max(y)) / 2
    sdf = dt(np.uint8(mask.numpy()))                                                       # :162-164
    sdf = sdf / sdf.max() if sdf.max() > 0 else sdf
    sdf = torch.tensor(sdf)
    all_data = torch.cat((image, sdf.unsqueeze(0), mask.unsqueeze(0)))                     # :165
    top, left, height, width = params
    all_data = all_data[:, top:top + height, left:left + width]                            # :169 (transforms.functional.crop)
    image = all_data[0:3]
    sdf = all_data[3:4]
    mask = all_data[-1]
    image = resize(image, (S, S))                                                          # :174
    mask = resize(mask.unsqueeze(0), (S, S), nearest=True)[0]                              # :175
    sdf = resize(sdf, (S, S))[0]                                                           # :176
    crop_center_y = (obj_y_center - top) * (S / height)                                    # :180-182
    crop_center_x = (obj_x_center - left) * (S / width)
    object_center = torch.tensor([crop_center_x, crop_center_y])
    if use_bg_sdf:                                                                         # :191-197
        bg_mask = torch.where(mask == 0, 1, 0)
        bg_sdf = dt(np.uint8(bg_mask.numpy()))
        bg_sdf = bg_sdf / bg_sdf.max() if bg_sdf.max() > 0 else bg_sdf
        sdf = sdf + torch.tensor(bg_sdf) * (-1)
    H, W = mask.shape
    xv, yv = torch.meshgrid([torch.arange(H), torch.arange(W)], indexing="ij")             # :200-207
    grid = torch.stack((xv, yv), 2).float().permute(2, 0, 1)
    ocf = grid - torch.tensor([object_center[1], object_center[0]]).unsqueeze(1).unsqueeze(1)
    ocf = F.normalize(ocf, dim=0)
    center_field = torch.zeros_like(grid) + torch.where(mask > 0, 1, 0) * ocf
    center_field = F.normalize(center_field, dim=0)
    return image, {"center_field": center_field, "saliency_mask": torch.where(mask > 0, 1, 0).float(), "instance_mask": mask.float(),
                   "object_center": object_center, "sdf": sdf}
