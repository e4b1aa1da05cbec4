"""Per-image time of the object-reasoning pipeline (unmore_amd.object_discovery.Object_Discovery.discover_image = the body of the
reference's main_object_discovery, object_reasoning.py:619-662) on a synthetic 640 x 480 scene.

The networks are REAL (ObjectnessNet with the reference's dpt_large backbone, Binary_Classifier; random weights -- there are no
checkpoints here) and do all their work; what they return is replaced by the stand-ins of tests/discovery_stubs.py (fields read back out
of the crop), so the CONTROL FLOW -- how many boxes exist, split, survive each boundary round -- is that of a scene with objects in it
instead of whatever random weights would say.  Three arms, same boxes out:
   reference_flow : every surviving box evaluated in every round, both heads every time (what object_reasoning.py does, on these kernels)
   sdf_only       : + the boundary rounds evaluate the boundary-distance head only
   sdf_only+carry : + boxes that are fixed points of a round are carried
   +batch200      : + 200 crops per net call in the boundary rounds instead of 50
   default        : + boxes with the same integer corners (= the same crop) are evaluated once per IMAGE (shared within a round,
                    remembered across rounds), and centre reasoning goes through the pipelined, certificate-driven sweep
python tools/discovery_bench.py [fp32|bf16] [backbone=dpt_large] [arm prefix, e.g. default]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from argparse import Namespace
from discovery_stubs import FieldsFromCrop, ObjectFraction
from unmore_amd import synth
from unmore_amd.binary_classifier import Binary_Classifier
from unmore_amd.object_discovery import Object_Discovery
from unmore_amd.objectness_net import ObjectnessNet

dt = sys.argv[1] if len(sys.argv) > 1 else "fp32"
backbone = sys.argv[2] if len(sys.argv) > 2 else "dpt_large"
only = sys.argv[3] if len(sys.argv) > 3 else None          # e.g. "default": that arm alone (for a profile)
dev = "cuda:0"
args = Namespace(use_bg_sdf=True, sdf_activation="tanh")
torch.manual_seed(0)
net = ObjectnessNet(dev, 128, backbone, args).to(dev).eval()
net.set_compute_dtype(torch.float32 if dt == "fp32" else torch.bfloat16)
clf = Binary_Classifier(dev, 128, args).to(dev).eval()
for m in (net, clf):
    for p in m.parameters():
        p.requires_grad = False


class RealWorkStubAnswer(torch.nn.Module):
    _HEAD_OF = ObjectnessNet._HEAD_OF          # Object_Discovery asks for single heads when the model can do that
    compute_dtype = property(lambda self: net.compute_dtype)     # ... and sweeps centre reasoning over three streams (fp32: certified precision)

    def __init__(self, honour_heads):
        super().__init__()
        self.honour, self.stub, self.calls, self.crops = honour_heads, FieldsFromCrop(), 0, 0

    def get_prediction(self, images, heads=None):
        net.get_prediction(images, heads=(heads if self.honour else None))
        self.calls += 1
        self.crops += len(images)
        out = self.stub(images)
        return out if heads is None else {k: out[k] for k in heads}

    forward = get_prediction


class RealClassifierStubAnswer(torch.nn.Module):
    def forward(self, images):
        clf(images)
        return ObjectFraction()(images)


H, W = 480, 640
image = torch.from_numpy(synth.reasoning_scene(H, W, seed=2, n_objects=6)).to(dev)
rows, ref_boxes = [], None
for name, honour, carry, nb, share in (("reference_flow", False, False, 50, False), ("sdf_only", True, False, 50, False),
                                       ("sdf_only+carry", True, True, 50, False), ("sdf_only+carry+batch200", True, True, 200, False),
                                       ("default (+every distinct crop once per image, centre reasoning swept)", True, True, 200, True)):
    if only is not None and not name.startswith(only):
        continue
    model = RealWorkStubAnswer(honour)
    od = Object_Discovery(Namespace(), dev, objectness_model=model, binary_classifier_model=RealClassifierStubAnswer())
    od.carry_fixed_points, od.boundary_batch, od.share_equal_crops = carry, nb, share
    od.pipelined_center_sweep = share          # (the last arm = every default)
    od.remember_crops = share
    phases = {}

    def timed(name, fn):
        def run(*a, **k):
            torch.cuda.synchronize()
            t = time.perf_counter()
            r = fn(*a, **k)
            torch.cuda.synchronize()
            phases[name] = phases.get(name, 0.0) + time.perf_counter() - t
            return r
        return run
    for meth in ("existence_checking", "center_reasoning", "boundary_reasoning"):
        setattr(od, meth, timed(meth, getattr(od, meth)))
    boxes = od.discover_image(image)          # warm-up: packs weights, captures the graphs of the recurring batch shapes
    torch.cuda.synchronize()
    model.calls = model.crops = 0
    phases.clear()
    t0 = time.perf_counter()
    boxes = od.discover_image(image)
    torch.cuda.synchronize()
    dtm = time.perf_counter() - t0
    if ref_boxes is None:
        ref_boxes = boxes
    same = boxes is not None and ref_boxes is not None and boxes.shape == ref_boxes.shape and bool(torch.equal(boxes, ref_boxes))
    rows.append({"arm": name, "backbone": backbone, "dtype": dt, "image": [H, W], "seconds_per_image": round(dtm, 3), "net_calls": model.calls,
                 "crops_through_the_net": model.crops, "boundary_rounds": od.stats.get("boundary_rounds"), "boundary_crops": od.stats.get("boundary_crops"), "boundary_distinct_crops": od.stats.get("boundary_distinct_crops"),
                 "boxes_out": None if boxes is None else len(boxes), "same_boxes_as_reference_flow": same,
                 "seconds_by_phase": {k: round(v, 3) for k, v in phases.items()}})
    print(json.dumps(rows[-1]), flush=True)

# ---- several images: one at a time against their boundary rounds in lock-step (discover_images), every default on
if only is None or only.startswith("lock"):
    scenes = [torch.from_numpy(synth.reasoning_scene(H, W, seed=sd, n_objects=6)).to(dev) for sd in (2, 3, 4, 5)]
    model = RealWorkStubAnswer(True)
    od = Object_Discovery(Namespace(), dev, objectness_model=model, binary_classifier_model=RealClassifierStubAnswer())
    for name, fn in (("4 images, one at a time", lambda: [od.discover_image(im) for im in scenes]),
                     ("4 images, boundary rounds in lock-step", lambda: od.discover_images(scenes))):
        fn()
        torch.cuda.synchronize()
        model.calls = model.crops = 0
        t0 = time.perf_counter()
        boxes = fn()
        torch.cuda.synchronize()
        dtm = time.perf_counter() - t0
        print(json.dumps({"arm": name, "backbone": backbone, "dtype": dt, "image": [H, W], "seconds_per_image": round(dtm / len(scenes), 3),
                          "net_calls": model.calls, "crops_through_the_net": model.crops, "boxes_out": [None if b is None else len(b) for b in boxes]}), flush=True)
