"""cProfile of the HOST side of the reference recipe's train step (eager): where the ~19 us per launch go."""
import cProfile
import pstats
import sys
from argparse import Namespace

import torch

sys.path.insert(0, ".")
from unmore_amd import synth  # noqa: E402
from unmore_amd.objectness_net import ObjectnessNet  # noqa: E402
from unmore_amd.trainer import TrainStep  # noqa: E402

dt = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else torch.float32
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = ObjectnessNet(dev, 128, "dpt_large", Namespace(use_bg_sdf=True, sdf_activation="tanh")).to(dev)
net.set_compute_dtype(dt)
step = TrainStep(net, lr=1e-4)
batch = [torch.from_numpy(x).to(dev) for x in synth.make_batch(20, 128, 128, seed=0)]
for _ in range(3):
    step.step(*batch)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step.step(*batch)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
