"""Host time to ENQUEUE one cfg2 train step vs the step's GPU time -- and the same launch list on a tiny input (dpt_base, 64x64, one
image: every kernel takes microseconds, so its enqueue time is the host's own work per step: Python + ctypes + hipLaunchKernel).
If the tiny step's host time is far below the cfg2 figure, the difference is queue back-pressure: the launch call blocks while the
stream's queue is full of 30-ms kernels, i.e. a waiting thread, not work (host_enqueue_ms_per_step)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from argparse import Namespace
from unmore_amd import synth
from unmore_amd.objectness_net import ObjectnessNet
from unmore_amd.trainer import TrainStep
dev = torch.device("cuda:0")
for size, batch in ((64, 1), (384, 64)):
    torch.manual_seed(0)
    net = ObjectnessNet(dev, size, "dpt_base", Namespace(use_bg_sdf=True, sdf_activation="tanh")).to(dev)
    net.set_compute_dtype(torch.bfloat16); net.train()
    step = TrainStep(net, lr=1e-4).set_graph_mode("off")
    os.environ["UMR_WGRAD_STREAM"] = "0"
    img, cf, sdf, sal = (torch.from_numpy(x).to(dev) for x in synth.make_batch(batch, size, size, seed=0))
    for _ in range(3): step.step(img, cf, sdf, sal)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); step.step(img, cf, sdf, sal); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        ts.append((t1 - t0, t2 - t0))
    print(f"dpt_base {size}x{size} batch {batch}: host enqueue ms / full step ms (queue drained before every step):",
          [(round(a * 1e3, 1), round(b * 1e3, 1)) for a, b in ts], flush=True)
    del step, net
    torch.cuda.empty_cache()
