"""how many of the distinct crops a boundary round evaluates were already evaluated in an EARLIER round of the same image?
(stub networks: control flow only)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from argparse import Namespace
from discovery_stubs import FieldsFromCrop, ObjectFraction
from unmore_amd import synth, reasoning
from unmore_amd.object_discovery import Object_Discovery
dev = "cuda:0"
for (H, W, seed, nobj) in ((480, 640, 2, 6), (240, 320, 0, 4), (200, 288, 5, 6)):
    image = torch.from_numpy(synth.reasoning_scene(H, W, seed, nobj)).to(dev)
    od = Object_Discovery(Namespace(), dev, objectness_model=FieldsFromCrop(), binary_classifier_model=ObjectFraction())
    seen, per_round = set(), []
    orig = reasoning.crop_resize
    def spy(image_, boxes, size=128):
        if boxes.is_cuda and boxes.dtype == torch.float64 and spy.on:
            ks = [tuple(int(v) for v in r) for r in boxes.cpu().tolist()]
            new = [k for k in ks if k not in seen]
            per_round.append((len(ks), len(new)))
            seen.update(ks)
        return orig(image_, boxes, size)
    spy.on = False
    reasoning.crop_resize = spy
    import unmore_amd.object_discovery as odm
    _br = od.boundary_reasoning
    def br(*a, **k):
        spy.on = True
        r = _br(*a, **k)
        spy.on = False
        return r
    od.boundary_reasoning = br
    od.discover_image(image)
    reasoning.crop_resize = orig
    tot = sum(a for a, _ in per_round); new = sum(b for _, b in per_round)
    print(f"{H}x{W}: distinct crops summed over rounds {tot}, never seen before {new} ({100*new/max(tot,1):.0f} %); per round (evaluated/new) first 12: {per_round[:12]} ... last 5: {per_round[-5:]}")
