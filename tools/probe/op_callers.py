"""Per call site inside unmore_amd/: how often one train step of the reference recipe (dpt_large 128x128, batch 20) calls each
`ops` function (one call = at least one kernel launch).   python tools/probe/op_callers.py [bf16|fp32] [op-name-filter]"""
import collections
import sys
import traceback
import types

import torch

sys.path.insert(0, ".")
from argparse import Namespace

from unmore_amd import ops, synth
from unmore_amd.objectness_net import ObjectnessNet
from unmore_amd.trainer import TrainStep

dtype = torch.float32 if (len(sys.argv) > 1 and sys.argv[1] == "fp32") else torch.bfloat16
flt = sys.argv[2] if len(sys.argv) > 2 else ""
net = ObjectnessNet("cuda:0", 128, "dpt_large", Namespace(use_bg_sdf=True, sdf_activation="tanh")).to("cuda:0")
net.set_compute_dtype(dtype)
step = TrainStep(net, lr=1e-4).set_graph_mode("off")
batch = tuple(torch.from_numpy(a).cuda() for a in synth.make_batch(20, 128, 128, seed=1))
for _ in range(3):
    step.step(*batch)
torch.cuda.synchronize()

per_op = collections.Counter()
per_site = collections.Counter()


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "unmore_amd" in fr.filename and not fr.filename.endswith("ops.py"):
            return f"{fr.filename.split('unmore_amd/')[-1]}:{fr.lineno} {fr.line.strip()[:100]}"
    return "?"


for name, fn in list(vars(ops).items()):
    if isinstance(fn, types.FunctionType) and not name.startswith("_") and name not in ("set_f32_mode", "get_f32_mode", "set_kernel_timer"):
        def make(name, fn):
            def f(*a, **k):
                per_op[name] += 1
                if flt and flt in name:
                    per_site[(name, site())] += 1
                return fn(*a, **k)
            return f
        setattr(ops, name, make(name, fn))
step.step(*batch)
torch.cuda.synchronize()
print("calls per step, by op:")
for name, n in per_op.most_common():
    print(f"{n:6d}  {name}")
if flt:
    print(f"\ncall sites of ops matching '{flt}':")
    for (name, where), n in per_site.most_common(40):
        print(f"{n:5d}  {name:14s} {where}")
