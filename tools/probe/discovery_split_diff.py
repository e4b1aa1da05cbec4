"""which second-level split boxes differ from the reference fixture (scene b), and by how much"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from argparse import Namespace
from discovery_stubs import FieldsFromCrop, ObjectFraction
from unmore_amd import synth, reasoning
from unmore_amd.object_discovery import Object_Discovery
G = np.load("tests/golden/discovery.npz")
od = Object_Discovery(Namespace(), "cuda:0", objectness_model=FieldsFromCrop(), binary_classifier_model=ObjectFraction())
image = torch.from_numpy(synth.reasoning_scene(200, 288, 5, 6)).cuda()
split1 = torch.from_numpy(G["b_split1"]).cuda()
ex = torch.from_numpy(G["b_existence1"])
props = split1[(ex >= 0.1).cuda()]
sdf, cen = od.get_prediction_with_proposals(props, image)
mx, am, sc = reasoning.center_peaks(sdf, cen, return_scores=True)
fail = mx > 0.009
cr = od.center_reasoning(image, props)
got, ref = cr["splited_new_proposals"].cpu().numpy(), G["b_split2"]
bad = np.nonzero((got != ref).any(1))[0]
print("rows differing", len(bad), "of", len(ref), "boxes:", sorted(set(bad // 4)))
fi = torch.nonzero(fail).flatten()
for b in sorted(set(bad // 4)):
    i = int(fi[b])
    s = sc[i].flatten()
    top = torch.topk(s, 4)
    print("box", b, props[i].tolist(), "got cut", got[4 * b, 2], got[4 * b + 2, 3], "ref cut", ref[4 * b, 2], ref[4 * b + 2, 3])
    print("   top scores", [f"{v:.12f}" for v in top.values.tolist()], "at", [(int(j) // 128, int(j) % 128) for j in top.indices])
