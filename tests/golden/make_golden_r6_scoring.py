"""Round-6 golden fixture for the second inference-path caller: the reference's `Object_Scoring` (object_scoring.py:43-272), made by
running the REFERENCE's own code on the CPU.

Run in the build container only (needs /root/reference; never shipped):
    python tests/golden/make_golden_r6_scoring.py      (after make_golden_r6_discovery.py: it scores that fixture's discovered boxes)

What executes verbatim from the reference: `get_prediction_with_proposals` (:112-157, bound to an instance made with `__new__`) and the
source LINES :182-235 and :238-245 of `main_object_scoring` (the per-image body has no callable entry point: it sits inside the loop
over a COCO dataset), read from the reference file at run time, dedented and exec'd in a namespace that supplies `self`, `image`,
`raw_proposals`, `torch`, `math`, `np`, `transforms`, `torchvision` -- nothing of that text is stored.
The two networks are the stand-ins of tests/discovery_stubs.py.
What cannot execute (absent from this image; UNPINNED boundaries, restated from the libraries' documented behaviour):
  * torchvision `transforms.Resize(size, BILINEAR)` on a tensor (0.14.1, README.md:25): F.interpolate(bilinear, align_corners=False,
    no antialias) in float32; for an INTEGER tensor -- the binary masks of :191,:207 are int64 -- the result is rounded (torch.round,
    half to even) and cast back (torchvision/transforms/functional_tensor.py: _cast_squeeze_in / _cast_squeeze_out);
  * pycocotools `mask.toBbox(encode(mask))` (:160-165): [x_min, y_min, width, height] of the mask's non-zero pixels, zeros if empty;
  * torchvision.ops.nms (:238): oracle/objectness_oracle.py::nms.
Only DATA is written."""
import math
import os
import sys
import textwrap

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import make_golden_r2 as g2          # noqa: E402
from unmore_amd import synth          # noqa: E402
from oracle import objectness_oracle as orc   # noqa: E402
from discovery_stubs import FieldsFromCrop, ObjectFraction   # noqa: E402


class ResizeTensor:
    def __init__(self, size, interpolation=None):
        self.size = size

    def __call__(self, img):
        out = F.interpolate(img.unsqueeze(0).to(torch.float32), size=self.size, mode="bilinear", align_corners=False).squeeze(0)
        if not img.is_floating_point():
            out = torch.round(out).to(img.dtype)
        return out


def to_bbox(binary_mask):
    ys, xs = np.nonzero(binary_mask)
    if len(ys) == 0:
        return [0.0, 0.0, 0.0, 0.0]
    return [float(xs.min()), float(ys.min()), float(xs.max() - xs.min() + 1), float(ys.max() - ys.min() + 1)]


def main():
    g2.install_placeholders()
    for name in ("skimage.measure", "torchvision.ops"):          # import-only for this path (inert: calling one raises)
        if name not in sys.modules:
            sys.modules[name] = g2._Inert(name)
    sys.modules["skimage"].measure = sys.modules["skimage.measure"]
    tv = sys.modules["torchvision"]
    tv.ops = sys.modules["torchvision.ops"]
    tv.transforms.Resize = ResizeTensor
    tv.ops.nms = lambda boxes, scores, iou_threshold: torch.from_numpy(orc.nms(boxes.numpy(), scores.numpy(), iou_threshold))
    sys.path.insert(0, REF)
    import object_scoring as osc
    OS = osc.Object_Scoring
    self = OS.__new__(OS)
    self.device = torch.device("cpu")
    self.objectness_model, self.binary_classifier_model = FieldsFromCrop(), ObjectFraction()
    self.binary_mask_to_tight_bbox_coco_style = staticmethod(to_bbox).__func__
    with open(os.path.join(REF, "object_scoring.py")) as f:
        lines = f.readlines()
    body = textwrap.dedent("".join(lines[181:235]))          # 1-based :182-235
    tail = textwrap.dedent("".join(lines[237:245]))          # :238-245
    assert body.lstrip().startswith("predictions = self.get_prediction_with_proposals(image, raw_proposals)") and "tight_bboxes = torch.FloatTensor(tight_bboxes)" in body
    assert tail.lstrip().startswith("nms_indexes = torchvision.ops.nms(") and "mask_scores = " in tail
    D = np.load(os.path.join(HERE, "discovery.npz"))
    save = {}
    for tag, (H, W, seed, nobj) in {"a": (240, 320, 0, 4), "b": (200, 288, 5, 6)}.items():
        image = torch.from_numpy(synth.reasoning_scene(H, W, seed, nobj))
        raw = D[f"{tag}_final_boxes"][D[f"{tag}_final_labels"] == 1].astype(np.float64)
        raw = np.concatenate([raw, np.array([[0.0, 0.0, W, H], [3.2, 4.7, 9.1, 8.9]])], axis=0)   # + the whole image and a tiny box in the background
        ns = dict(self=self, image=image, raw_proposals=raw.tolist(), torch=torch, math=math, np=np, transforms=tv.transforms, torchvision=tv)
        exec(compile(body, "object_scoring.py:182-235", "exec"), ns)
        exec(compile(tail, "object_scoring.py:238-245", "exec"), ns)
        save[f"{tag}_raw_proposals"] = raw
        save[f"{tag}_max_center"] = ns["max_center_fields_norms"].numpy()
        save[f"{tag}_max_boundary"] = ns["max_boundary_distance_values"].numpy()
        save[f"{tag}_existence"] = ns["pred_existence_scores"].numpy()
        save[f"{tag}_tight"] = ns["tight_bboxes"].numpy()
        save[f"{tag}_union_area"] = ns["resized_union_binary_masks"].sum(1).sum(1).numpy()
        save[f"{tag}_nms"] = ns["nms_indexes"].numpy()
        save[f"{tag}_mask_scores"] = ns["mask_scores"]
        fm = ns["final_binary_masks"].numpy().astype(np.uint8)
        save[f"{tag}_final_masks_packed"] = np.packbits(fm, axis=None)
        save[f"{tag}_final_masks_shape"] = np.array(fm.shape)
        sc = ns["existence_scores"] * ns["center_scores"] * ns["boundary_scores"] * np.power(ns["mask_scores"], 0.25)    # :255, per box
        save[f"{tag}_score"] = sc
        print(tag, len(raw), "proposals ->", len(ns["nms_indexes"]), "after NMS; score range", float(sc.min()), float(sc.max()), "dtype", sc.dtype,
              "empty masks", int((save[f"{tag}_union_area"] == 0).sum()))
    path = os.path.join(HERE, "scoring.npz")
    np.savez_compressed(path, **save)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
