"""Device-side glue of unMORE's object reasoning around the ObjectnessNet calls:
proposal crop + resize, centre-field peak picking, boundary-field box deltas.  Each function replaces a Python /
PyTorch block of the reference's `Object_Discovery` (object_reasoning.py) with one kernel launch."""
import ctypes
import math

import numpy as np
import torch

from . import _lib as L
from .ops import _p, _stream, _need_gpu


def crop_resize(image, boxes, size=128):
    """object_reasoning.py:311-323 -- for every box: floor/ceil the corners, crop `image[:, y1:y2, x1:x2]`, resize to
    (size, size) with bilinear interpolation (torchvision tensor Resize, no antialias).  image: [3,H,W] f32 on the GPU,
    boxes: [N,4] (x1,y1,x2,y2) any float/int tensor, on the host or the GPU.  Returns ([N,3,size,size] f32, on_edge_flags [N,4] bool on
    the boxes' device)."""
    _need_gpu(image)
    assert image.dim() == 3 and image.shape[0] == 3 and image.dtype == torch.float32
    image = image.contiguous()
    H, W = image.shape[1], image.shape[2]
    # corners as the reference takes them (int(math.floor(x1)) ... int(math.ceil(y2))), where the boxes live: boxes on the GPU stay
    # there -- the boundary-reasoning rounds (object_discovery.py) feed each round's boxes to the next without a host round trip
    b = boxes.detach().to(torch.float64)
    ib = torch.stack([torch.floor(b[:, 0]), torch.floor(b[:, 1]), torch.ceil(b[:, 2]), torch.ceil(b[:, 3])], 1).to(torch.int32)
    # python slicing semantics of image[:, y1:y2, x1:x2]: negative starts would wrap; the reference clips boxes beforehand
    ib[:, 0].clamp_(0, W); ib[:, 2].clamp_(0, W); ib[:, 1].clamp_(0, H); ib[:, 3].clamp_(0, H)
    on_edge = torch.stack([ib[:, 0] == 0, ib[:, 1] == 0, ib[:, 2] == W, ib[:, 3] == H], 1)
    N = ib.shape[0]
    out = torch.empty((N, 3, size, size), dtype=torch.float32, device=image.device)
    ibd = ib.to(image.device)
    L.check(L.lib().umr_crop_resize_bilinear(_p(image), _p(ibd), _p(out), N, H, W, size, _stream()), "umr_crop_resize_bilinear")
    return out, on_edge


_filter_cache = {}


def _anti_center_filter(device):
    """normalize((2-i, 2-j)) per tap in float32 (F.normalize), promoted to float64 (object_reasoning.py:368-373)."""
    f = _filter_cache.get(device)
    if f is None:
        g = np.zeros((2, 5, 5), np.float32)
        for i in range(5):
            for j in range(5):
                v = np.array([2 - i, 2 - j], np.float32)
                n = np.float32(np.sqrt(np.float32(v[0] * v[0] + v[1] * v[1])))
                g[:, i, j] = v / max(n, np.float32(1e-12))
        f = torch.from_numpy(g.astype(np.float64)).to(device)
        _filter_cache[device] = f
    return f


def center_peaks(sdf_maps, center_fields, border=10, erode_kernel=9, erode_rounds=3, return_scores=False):
    """object_reasoning.py:528-550.  sdf_maps [B,H,W], center_fields [B,2,H,W] (f32, GPU).
    Returns (max score [B] f64, flat argmax [B] int64[, score maps [B,H,W] f64])."""
    _need_gpu(sdf_maps, center_fields)
    sdf = sdf_maps.contiguous().float()
    cen = center_fields.contiguous().float()
    B, H, W = sdf.shape
    mx = torch.empty(B, dtype=torch.float64, device=sdf.device)
    am = torch.empty(B, dtype=torch.int64, device=sdf.device)
    sc = torch.empty((B, H, W), dtype=torch.float64, device=sdf.device) if return_scores else None
    L.check(L.lib().umr_center_peaks(_p(sdf), _p(cen), _p(_anti_center_filter(sdf.device)), _p(sc), _p(mx), _p(am), B, H, W, border,
                                     erode_kernel, erode_rounds, _stream()), "umr_center_peaks")
    return (mx, am, sc) if return_scores else (mx, am)


def center_peaks_certified(sdf_maps, center_fields, eps, singular_threshold=0.009, border=10, erode_kernel=9, erode_rounds=3):
    """center_peaks plus, per map, whether its flat argmax is provably what ANY fields within `eps` (max-norm) of these would give
    (include/umr.h, umr_center_peaks_certified; singular_threshold: object_reasoning.py:541).
    Returns (max score [B] f64, flat argmax [B] int64, certified [B] bool)."""
    _need_gpu(sdf_maps, center_fields)
    sdf = sdf_maps.contiguous().float()
    cen = center_fields.contiguous().float()
    B, H, W = sdf.shape
    mx = torch.empty(B, dtype=torch.float64, device=sdf.device)
    am = torch.empty(B, dtype=torch.int64, device=sdf.device)
    ce = torch.empty(B, dtype=torch.int32, device=sdf.device)
    L.check(L.lib().umr_center_peaks_certified(_p(sdf), _p(cen), _p(_anti_center_filter(sdf.device)), _p(mx), _p(am), _p(ce), B, H, W, border,
                                               erode_kernel, erode_rounds, float(eps), float(singular_threshold), _stream()),
            "umr_center_peaks_certified")
    return mx, am, ce != 0


def update_bbox_with_boundary_fields(sdf_maps):
    """object_reasoning.py:139-174: (delta_x1, delta_y1, delta_x2, delta_y2), each [B]."""
    _need_gpu(sdf_maps)
    sdf = sdf_maps.contiguous().float()
    B, H, W = sdf.shape
    d = torch.empty((B, 4), dtype=torch.float32, device=sdf.device)
    L.check(L.lib().umr_boundary_deltas(_p(sdf), _p(d), B, H, W, _stream()), "umr_boundary_deltas")
    return d[:, 0], d[:, 1], d[:, 2], d[:, 3]


def mask_components(sdf_maps, center_fields, max_components=1024):
    """8-connected components of every map's union mask, in scipy.ndimage.label's order (object_reasoning.py:206-257).
    sdf_maps [B,S,S], center_fields [B,2,S,S] f32 on the GPU.  Returns (counts [B] int32, boxes [B,max_components,4] int32 [x1,y1,x2,y2),
    zeros beyond a map's count) on the device."""
    _need_gpu(sdf_maps, center_fields)
    sdf = sdf_maps.contiguous().float()
    cen = center_fields.contiguous().float()
    B, S = sdf.shape[0], sdf.shape[-1]
    counts = torch.empty(B, dtype=torch.int32, device=sdf.device)
    boxes = torch.empty((B, max_components, 4), dtype=torch.int32, device=sdf.device)
    if B:
        L.check(L.lib().umr_mask_components(_p(sdf), _p(cen), B, S, max_components, _p(counts), _p(boxes), _stream()), "umr_mask_components")
    return counts, boxes


def nms(boxes, scores, iou_threshold):
    """torchvision.ops.nms as object_reasoning.py:661 uses it: indices of the kept boxes, by descending score; equal scores keep their
    input order (the reference passes its labels -- all ones -- as scores; torchvision leaves the order of ties to its sort).
    boxes [N,4] f32 (x1,y1,x2,y2), scores [N], on the GPU.  Returns an int64 tensor on the GPU (one host sync for the kept count)."""
    _need_gpu(boxes)
    n = boxes.shape[0]
    if n == 0:
        return torch.empty(0, dtype=torch.int64, device=boxes.device)
    b = boxes.detach().to(torch.float32).contiguous()
    order = torch.sort(scores.detach().to(boxes.device), descending=True, stable=True).indices.contiguous()
    ws_bytes = int(L.lib().umr_nms_workspace(n))
    ws = torch.empty(ws_bytes // 8, dtype=torch.int64, device=boxes.device)
    keep = torch.empty(n, dtype=torch.int64, device=boxes.device)
    cnt = torch.zeros(1, dtype=torch.int32, device=boxes.device)
    L.check(L.lib().umr_nms(_p(b), _p(order), n, float(iou_threshold), _p(ws), ws_bytes, _p(keep), _p(cnt), _stream()), "umr_nms")
    return keep[:int(cnt.item())]


def existence_checking(binary_classifier_model, image, proposals, num_img_per_batch=128):
    """object_reasoning.py:491-523: crop every proposal, bilinear-resize to 128x128, score the crops
    with the existence classifier in batches of 128.  image [3,H,W] f32 on the GPU; proposals [N,4] (x1,y1,x2,y2).
    Returns the reference's dict: {'existence_scores': [N] f32 on the CPU}."""
    scores = []
    for i in range(0, len(proposals), num_img_per_batch):
        crops, _ = crop_resize(image, proposals[i:i + num_img_per_batch], 128)
        with torch.no_grad():
            scores.append(binary_classifier_model(crops.to(torch.float32)))
    class_scores = torch.cat(scores, dim=0).cpu()
    return {"existence_scores": class_scores.squeeze(1)}


def get_prediction_with_proposals(objectness_model, binary_classifier_model, image, proposals, num_img_per_batch=50):
    """object_scoring.py:111-153: for every proposal box the 128x128 crop (floor / ceil corners, bilinear, no antialias), the
    objectness net's boundary-distance and centre fields and the existence score, in batches of 50.
    image [3,H,W] f32 on the GPU; proposals [N,4] (x1,y1,x2,y2).  Returns the reference's dict -- 'pred_boundary_fields'
    [N,128,128], 'pred_center_fields' [N,2,128,128], 'pred_existence_scores' [N] -- on the model's device, plus
    'on_edge_flags' [N,4] (computed by the reference loop, :126-127, and dropped from its return value)."""
    props = torch.as_tensor(proposals, dtype=torch.float64)
    sdf, cen, cls, edge = [], [], [], []
    for i in range(0, len(props), num_img_per_batch):
        crops, on_edge = crop_resize(image, props[i:i + num_img_per_batch], 128)
        with torch.no_grad():
            pred = objectness_model(crops.to(torch.float32))
            cls.append(binary_classifier_model(crops.to(torch.float32)))
        sdf.append(pred["sdf_maps"].squeeze(1))
        cen.append(pred["center_fields"])
        edge.append(on_edge)
    return {"pred_boundary_fields": torch.cat(sdf, dim=0), "pred_center_fields": torch.cat(cen, dim=0),
            "pred_existence_scores": torch.cat(cls, dim=0).squeeze(1), "on_edge_flags": torch.cat(edge, dim=0)}


_sweep_streams = {}


CERT_EPS = 2e-4     # the field perturbation the certificate covers: twice the 1e-4 parity contract (tests/golden/make_golden_r2.py: CERT_EPS)


def _sweep_pass(objectness_model, image, props, num_img_per_batch, streams, cur, certify):
    outs = []
    for s in streams:
        s.wait_stream(cur)
    for bi, i in enumerate(range(0, len(props), num_img_per_batch)):
        st = streams[bi % len(streams)]
        with torch.cuda.stream(st):
            crops, _ = crop_resize(image, props[i:i + num_img_per_batch], 128)
            with torch.no_grad():
                pred = objectness_model.get_prediction(crops)
            sdf = pred["sdf_maps"].squeeze(1)
            if certify:
                mx, am, ce = center_peaks_certified(sdf, pred["center_fields"], CERT_EPS)
            else:
                mx, am = center_peaks(sdf, pred["center_fields"])
                ce = torch.ones_like(am, dtype=torch.bool)
            d = torch.stack(update_bbox_with_boundary_fields(sdf), 1)
            # the results are consumed (torch.cat) on the caller's stream after cur.wait_stream(st): tell the allocator, so the
            # blocks are not handed to the next batch of this side stream before that read has run
            for t in (mx, am, d, ce):
                t.record_stream(cur)
            outs.append((mx, am, d, ce))
    for s in streams:
        cur.wait_stream(s)
    return tuple(torch.cat([o[k] for o in outs]) for k in range(4))


def sweep_proposals(objectness_model, image, proposals, num_img_per_batch=50, n_streams=3, precision="full", info=None):
    """The per-proposal part of `center_reasoning` + the first half of `optimize_one_image_single_round` for ALL proposals of an
    image (object_reasoning.py:301-337,528-550,139-174): crop + resize, ObjectnessNet maps, centre peaks, boundary box deltas,
    in the reference's batches of 50.  Batches are independent, so consecutive batches are enqueued on `n_streams` HIP streams
    in turn: one batch's chain of ~330 small dependent kernels overlaps the other's large head convolutions (results are
    identical to the sequential order -- every batch runs the same kernels on the same data).
    precision (fp32 compute mode only; bf16 nets ignore it):
      'full'      every proposal in the library's current f32 product mode (default: six-term fp32-grade products);
      'certified' certificate-driven (round 6): pass 1 runs every proposal with THREE-term products (ops.set_f32_mode('x3_fast'):
                  maps within ~4e-5 of the six-term ones, half the matrix work) and asks the device, per proposal, whether its peak
                  index is provably the one ANY fields within CERT_EPS = 2e-4 of these maps would give (center_peaks_certified); pass 2
                  re-runs only the proposals without that certificate in the full mode and takes their results from there.  Costs one
                  host synchronisation per image (the reference synchronises per proposal, object_reasoning.py:546-550).
    info: an optional dict that receives {'proposals', 'rerun', 'certified_in_pass1'}.
    Returns (max_values [N] f64, flat_argmax [N] i64, deltas [N,4] f32) on the device."""
    from . import ops
    _need_gpu(image)
    dev = image.device
    key = (dev.index, n_streams)
    if key not in _sweep_streams:
        _sweep_streams[key] = [torch.cuda.Stream(device=dev) for _ in range(n_streams)]
    streams = _sweep_streams[key]
    props = torch.as_tensor(proposals, dtype=torch.float64)
    cur = torch.cuda.current_stream(dev)
    certified_mode = (precision == "certified" and getattr(objectness_model, "compute_dtype", torch.float32) == torch.float32
                      and ops.get_f32_mode() == "x3")
    assert precision in ("full", "certified")
    if not certified_mode:
        mx, am, d, _ = _sweep_pass(objectness_model, image, props, num_img_per_batch, streams, cur, certify=False)
        if info is not None:
            info.update(proposals=len(props), rerun=0, certified_in_pass1=None)
        return mx, am, d
    prev = ops.set_f32_mode("x3_fast")
    try:
        mx, am, d, ce = _sweep_pass(objectness_model, image, props, num_img_per_batch, streams, cur, certify=True)
    finally:
        ops.set_f32_mode(prev)
    redo = torch.nonzero(~ce).flatten().cpu()          # the one host synchronisation of the image
    if info is not None:
        info.update(proposals=len(props), rerun=int(redo.numel()), certified_in_pass1=int(len(props) - redo.numel()))
    if redo.numel():
        mx2, am2, d2, _ = _sweep_pass(objectness_model, image, props[redo], num_img_per_batch, streams, cur, certify=False)
        idx = redo.to(dev)
        mx[idx], am[idx], d[idx] = mx2, am2, d2
    return mx, am, d
