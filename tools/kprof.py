"""Run ONE kernel shape a few times (for rocprofv3 --pmc passes): python tools/kprof.py {convnt|convtn|gemm1x1} [B]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from unmore_amd import ops, _lib as L

which = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
H = W = 384
dev = torch.device("cuda:0")
dt = torch.bfloat16
M = B * H * W
g = torch.Generator(device="cpu").manual_seed(0)
x512 = (torch.randn((B, H, W, 512), generator=g)).to(dev).to(dt)
if which == "convnt":
    w3 = (torch.randn((512, 4608), generator=g) * 0.02).to(dev).to(dt)
    out = torch.empty((M, 512), dtype=dt, device=dev)
    fn = lambda: ops.gemm_nt(x512, w3, torch.zeros(512, device=dev), conv=1, act=L.ACT_RELU, out=out)
elif which == "convtn":
    dy = torch.randn((M, 512), generator=g).to(dev).to(dt)
    fn = lambda: ops.gemm_tn(dy, x512, conv=1)
else:
    w1 = (torch.randn((1024, 512), generator=g) * 0.04).to(dev).to(dt)
    out2 = torch.empty((M, 1024), dtype=dt, device=dev)
    fn = lambda: ops.gemm_nt(x512.view(M, 512), w1, torch.zeros(1024, device=dev), act=L.ACT_RELU, out=out2)
for _ in range(3):
    fn()
torch.cuda.synchronize()
