"""Race hunt for the chain-of-graphs train step (graphs.StagedCaptured): the reference recipe's shape (dpt_large, 20 x 128^2), N steps,
the eager one-stream step and the default (captured, two lanes) step side by side from the same weights on the same batches -- every kernel
is deterministic, so the losses and the weights must stay BIT-IDENTICAL; a lane that read a buffer too early or too late would show
up as a difference at some step.   python tools/probe/staged_soak.py [steps=300] [bf16|fp32]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from argparse import Namespace
from unmore_amd import graphs, synth
from unmore_amd.objectness_net import ObjectnessNet
from unmore_amd.trainer import TrainStep

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dt = torch.float32 if (len(sys.argv) > 2 and sys.argv[2] == "fp32") else torch.bfloat16
dev = torch.device("cuda:0")
torch.manual_seed(0)
nets = []
for _ in range(2):
    torch.manual_seed(0)
    n = ObjectnessNet(dev, 128, "dpt_large", Namespace(use_bg_sdf=True, sdf_activation="tanh")).to(dev)
    n.set_compute_dtype(dt); n.train()
    nets.append(n)
nets[1].load_state_dict(nets[0].state_dict())
os.environ["UMR_WGRAD_STREAM"] = "0"
step_e = TrainStep(nets[0], lr=1e-4, lr_milestones=(100, 200), lr_gamma=0.5).set_graph_mode("off")
os.environ["UMR_WGRAD_STREAM"] = "auto"
step_g = TrainStep(nets[1], lr=1e-4, lr_milestones=(100, 200), lr_gamma=0.5)
from unmore_amd import engine
pool = []
for b in range(6):
    _, cf, sdf, sal = synth.make_batch(20, 128, 128, seed=500 + b)
    img = synth.blob_images(20, 128, 128, seed=500 + b)
    pool.append(tuple(torch.from_numpy(a).to(dev) for a in (img, cf, sdf, sal)))
bad = 0
t0 = time.perf_counter()
for it in range(steps):
    engine._WGRAD_STREAM = "0"
    le = step_e.step(*pool[it % 6])
    engine._WGRAD_STREAM = "auto"
    lg = step_g.step(*pool[it % 6])
    if not torch.equal(le, lg):
        bad += 1
        print(f"step {it + 1}: losses differ {le.tolist()} vs {lg.tolist()}", flush=True)
    if (it + 1) % 50 == 0:
        same = torch.equal(step_e.flat_p, step_g.flat_p)
        print(f"step {it + 1:4d}: loss {le[0].item():.4f}  weights bit-identical: {same}  replays {step_g.graph_replays}", flush=True)
        bad += 0 if same else 1
torch.cuda.synchronize()
caps = [v for v in step_g._graphs.values() if isinstance(v, graphs.CAPTURE_TYPES)]
print(f"{steps} steps in {time.perf_counter() - t0:.1f} s, {dt}; capture: {type(caps[0]).__name__ if caps else None}, "
      f"segments main/side: {[lane for lane, _ in caps[0].segments].count('main') if caps else 0}/{[lane for lane, _ in caps[0].segments].count('side') if caps else 0}; "
      f"differences: {bad}")
sys.exit(1 if bad else 0)
