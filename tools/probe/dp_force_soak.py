"""Race hunt for the data-parallel form of the chain-of-graphs train step THROUGH RCCL (a process group of one rank on the "nccl" backend,
parallel.BucketedAllReduce(force=True)): the reference recipe's shape (dpt_large, 20 x 128^2), N steps, the step without any exchange and the
data-parallel step (every bucket through ProcessGroupNCCL, the optimizer stage by stage two stages behind its bucket's all-reduce) side by
side from the same weights on the same batches.  A sum over one rank is the identity and every kernel is deterministic: losses and weights
must stay BIT-IDENTICAL; an optimizer launch that read a bucket before its exchange had landed (or a collective that started before the
weight gradients had) would show up as a difference at some step.   python tools/probe/dp_force_soak.py [steps=1000] [f32|bf16 wire] [auto|off graphs] [print every n] [seed]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29571")
import torch
import torch.distributed as dist
from argparse import Namespace
from unmore_amd import graphs, synth
from unmore_amd.objectness_net import ObjectnessNet
from unmore_amd.trainer import TrainStep

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
wire = torch.bfloat16 if (len(sys.argv) > 2 and sys.argv[2] == "bf16") else None
gmode = sys.argv[3] if len(sys.argv) > 3 else "auto"
every = int(sys.argv[4]) if len(sys.argv) > 4 else 100
seed = int(sys.argv[5]) if len(sys.argv) > 5 else 0
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
nets = []
for _ in range(2):
    torch.manual_seed(seed)
    n = ObjectnessNet(dev, 128, "dpt_large", Namespace(use_bg_sdf=True, sdf_activation="tanh")).to(dev)
    n.set_compute_dtype(torch.bfloat16); n.train()
    nets.append(n)
nets[1].load_state_dict(nets[0].state_dict())
os.environ["UMR_DP_FORCE"] = "0"
plain = TrainStep(nets[0], lr=1e-4, lr_milestones=(300, 600), lr_gamma=0.5).set_graph_mode(gmode)
os.environ["UMR_DP_FORCE"] = "1"
dp = TrainStep(nets[1], lr=1e-4, lr_milestones=(300, 600), lr_gamma=0.5, grad_wire_dtype=wire).set_graph_mode(gmode)
assert dp.comm.enabled and not plain.comm.enabled
pool = []
for b in range(6):
    _, cf, sdf, sal = synth.make_batch(20, 128, 128, seed=700 + 10 * seed + b)
    img = synth.blob_images(20, 128, 128, seed=700 + 10 * seed + b)
    pool.append(tuple(torch.from_numpy(a).to(dev) for a in (img, cf, sdf, sal)))
bad, worst = 0, 0.0
t0 = time.perf_counter()
mem0 = None
for it in range(steps):
    lp = plain.step(*pool[it % 6])
    ld = dp.step(*pool[it % 6])
    if wire is None and not torch.equal(lp, ld):
        bad += 1
        print(f"step {it + 1}: losses differ {lp.tolist()} vs {ld.tolist()}", flush=True)
    if (it + 1) % every == 0:
        torch.cuda.synchronize()
        mem = torch.cuda.memory_allocated() / 2 ** 30
        mem0 = mem0 or mem
        if wire is None:
            same = torch.equal(plain.flat_p, dp.flat_p)
            bad += 0 if same else 1
            print(f"step {it + 1:5d}: loss {lp[0].item():.4f}  weights bit-identical: {same}  replays {dp.graph_replays}  pending works {len(dp.comm._pending)}  "
                  f"allocated {mem:.2f} GiB", flush=True)
        else:
            rel = float((plain.flat_p - dp.flat_p).norm() / plain.flat_p.norm())
            worst = max(worst, rel)
            fin = bool(torch.isfinite(ld).all())
            bad += 0 if fin else 1
            print(f"step {it + 1:5d}: loss {lp[0].item():.4f} / {ld[0].item():.4f} (bf16 wire)  weights relative L2 apart {rel:.2e}  finite {fin}  "
                  f"allocated {mem:.2f} GiB", flush=True)
torch.cuda.synchronize()
mem = torch.cuda.memory_allocated() / 2 ** 30
caps = [v for v in dp._graphs.values() if isinstance(v, graphs.CAPTURE_TYPES)]
segs = [lane for lane, _ in caps[0].segments] if caps else []
print(f"{steps} steps of each in {time.perf_counter() - t0:.1f} s; wire {'bf16' if wire is not None else 'f32'}; chain segments: "
      f"{ {k: segs.count(k) for k in sorted(set(segs))} }; allocated memory {mem0:.2f} -> {mem:.2f} GiB; differences: {bad}"
      + (f"; worst weight distance {worst:.2e}" if wire is not None else ""))
dist.destroy_process_group()
sys.exit(1 if bad else 0)
