"""cfg5 (BASELINE.json configs[4]): inference-only object-reasoning sweep -- synthetic 640x480 images, the 1,225
anchors of object_reasoning.py:109-137 per image, crops resized to 128x128 in batches of 50, ObjectnessNet ViT-B/16
centre/boundary maps, centre peak picking -- all on the GPU.  Reports crops/s and images/s, and (fp32 mode) checks the
device peak indices of a sample against the CPU oracle's post-processing bit for bit.

    python tools/sweep_bench.py [--images 4] [--dtype fp32|bf16] [--backbone dpt_base]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def anchors(height, width):
    """proposal grid of object_reasoning.py:109-137 (restated: 5 scales x 3 aspect ratios + the full image)."""
    out = []
    for gs in (32, 64, 128, 256, 512):
        ys = np.arange(0, height, gs, dtype=int)
        xs = np.arange(0, width, gs, dtype=int)
        xc, yc = np.meshgrid(xs, ys)
        c = np.stack([xc.flatten(), yc.flatten(), xc.flatten(), yc.flatten()]).transpose().reshape(-1, 1, 4)
        base = np.array([[-gs, -gs, gs, gs], [-gs / 2, -gs, gs / 2, gs], [-gs, -gs / 2, gs, gs / 2]]).reshape(1, -1, 4)
        out.append((c + base).reshape(-1, 4))
    out = np.concatenate(out, 0).astype(np.float64)
    out[:, 0][out[:, 0] < 0] = 0
    out[:, 1][out[:, 1] < 0] = 0
    out[:, 2][out[:, 2] >= width] = width
    out[:, 3][out[:, 3] >= height] = height
    return np.concatenate((out, [[0, 0, width, height]]), 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=4)
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16"])
    ap.add_argument("--backbone", default="dpt_base")
    ap.add_argument("--check", type=int, default=100, help="crops whose peaks are checked against the CPU oracle")
    ap.add_argument("--sdf-head", default="factored", choices=["factored", "collapsed"], help="opt-in algebraic sdf head (DESIGN.md section 7)")
    a = ap.parse_args()
    from argparse import Namespace
    from unmore_amd import reasoning
    from unmore_amd.hashrng import uniform01
    from unmore_amd.objectness_net import ObjectnessNet

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = ObjectnessNet(dev, 128, a.backbone, Namespace(use_bg_sdf=True, sdf_activation="tanh")).to(dev).to(torch.float32).eval()
    for p in net.parameters():
        p.requires_grad = False
    net.set_compute_dtype(torch.bfloat16 if a.dtype == "bf16" else torch.float32)
    net.set_sdf_head_mode(a.sdf_head)
    H, W = 480, 640
    props = torch.from_numpy(anchors(H, W))
    assert props.shape[0] == 1225, props.shape
    imgs = [torch.from_numpy(uniform01(f"sweep:{i}", (3, H, W))).to(dev) for i in range(a.images)]

    def run_image(img, keep=None):
        res = []
        for b0 in range(0, props.shape[0], 50):
            crops, _ = reasoning.crop_resize(img, props[b0:b0 + 50], 128)
            with torch.no_grad():
                out = net.get_prediction(crops)
            sdf, cen = out["sdf_maps"].squeeze(1), out["center_fields"]
            mx, am = reasoning.center_peaks(sdf, cen)
            d = reasoning.update_bbox_with_boundary_fields(sdf)
            res.append((mx, am, d))
            if keep is not None and len(keep) * 50 < a.check:
                keep.append((sdf.cpu(), cen.cpu(), mx.cpu(), am.cpu()))
        return res

    run_image(imgs[0])  # warm-up
    torch.cuda.synchronize()
    keep = []
    t0 = time.perf_counter()
    for i, img in enumerate(imgs):
        run_image(img, keep if i == 0 else None)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ncrops = a.images * props.shape[0]
    res = {"metric": "object-reasoning sweep: crop+resize, ObjectnessNet maps, centre peaks, boundary deltas",
           "backbone": a.backbone, "dtype": a.dtype, "sdf_head": a.sdf_head, "images": a.images, "image_size": [W, H], "proposals_per_image": 1225,
           "crops_per_sec": ncrops / dt, "images_per_sec": a.images / dt, "est_minutes_for_5000_images": 5000 / (a.images / dt) / 60}
    # parity of the integer outputs: device peaks vs the CPU oracle on the same maps
    from oracle import objectness_oracle as orc
    checked = mism = 0
    for sdf, cen, mx, am in keep:
        _, m_r, a_r = orc.peak_pick(sdf, cen)
        checked += len(am)
        mism += int((am != a_r).sum())
    res["peaks_checked"] = checked
    res["peak_index_mismatches"] = mism
    print(json.dumps(res))


if __name__ == "__main__":
    main()
