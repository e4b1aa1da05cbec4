"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY (ground-truth synthesis).

Restates the label half of the reference's dataset item (datasets.py:158-159,171-222, branch without random crop).  Only
`tests/` may import this file; the product (`unmore_amd/`) never does.

Parity status: UNPINNED.  The reference computes the distance fields with `cv2.distanceTransform(u8, cv2.DIST_L2, 3)`
(datasets.py:176,186); OpenCV (opencv-python==4.10.0.84, requirements.txt) is absent from /root/reference and from this
image, the dataset class cannot be imported without it, and the reference holds no fixtures.  `distance_transform_3x3`
restates OpenCV's published algorithm for that call (imgproc/src/distransform.cpp `distanceTransform_3x3`: two-pass 3x3
chamfer in 16.16 fixed point with the documented DIST_L2 3x3 weights a = 0.955, b = 1.3693, a one-pixel border of
INT_MAX >> 2, output clamped to INT_MAX >> 2 and scaled by 2^-16).  The centre-field lines are the reference's own PyTorch
expressions restated one to one.
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
