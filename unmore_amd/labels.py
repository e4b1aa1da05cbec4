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
