"""Who issues the small device-to-device copies of a train step?  Counts, per call site inside unmore_amd/, the torch-level
calls that end in a copy kernel (copy_, clone, contiguous on a non-contiguous tensor, cat, fill_/zero_) during ONE step of the
reference recipe (dpt_large 128x128, batch 20).   python tools/probe/copy_callers.py [bf16|fp32]"""
import collections
import sys
import traceback

import torch

sys.path.insert(0, ".")
from argparse import Namespace

from unmore_amd import synth
from unmore_amd.objectness_net import ObjectnessNet
from unmore_amd.trainer import TrainStep

dtype = torch.float32 if (len(sys.argv) > 1 and sys.argv[1] == "fp32") else torch.bfloat16
net = ObjectnessNet("cuda:0", 128, "dpt_large", Namespace(use_bg_sdf=True, sdf_activation="tanh")).to("cuda:0")
net.set_compute_dtype(dtype)
step = TrainStep(net, lr=1e-4).set_graph_mode("off")
batch = tuple(torch.from_numpy(a).cuda() for a in synth.make_batch(20, 128, 128, seed=1))
for _ in range(3):
    step.step(*batch)
torch.cuda.synchronize()

counts = collections.Counter()


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "unmore_amd" in fr.filename:
            return f"{fr.filename.split('unmore_amd/')[-1]}:{fr.lineno} {fr.line.strip()[:90]}"
    return "?"


def wrap(obj, name, cond=lambda *a, **k: True):
    orig = getattr(obj, name)

    def f(*a, **k):
        if cond(*a, **k):
            counts[(name, site())] += 1
        return orig(*a, **k)
    setattr(obj, name, f)


wrap(torch.Tensor, "copy_")
wrap(torch.Tensor, "clone")
wrap(torch.Tensor, "contiguous", lambda t, *a, **k: not t.is_contiguous())
wrap(torch.Tensor, "zero_")
wrap(torch.Tensor, "fill_")
wrap(torch.Tensor, "to", lambda t, *a, **k: True)
wrap(torch, "cat")
wrap(torch, "zeros")
step.step(*batch)
torch.cuda.synchronize()
for (name, where), n in counts.most_common(40):
    print(f"{n:5d}  {name:11s} {where}")
