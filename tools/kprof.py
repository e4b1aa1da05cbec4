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
elif which == "convnt_masked":
    # the centre head's ReLU-masked 3x3 data gradient
    w3 = (torch.randn((512, 4608), generator=g) * 0.02).to(dev).to(dt)
    out = torch.empty((M, 512), dtype=dt, device=dev)
    h1 = torch.randn((M // 8, 512), generator=g).to(dt).to(dev).repeat(8, 1)
    fn = lambda: ops.gemm_nt(x512, w3, None, conv=1, aux=h1, mask_relu=True, out=out)
elif which == "convtn":
    dy = torch.randn((M, 512), generator=g).to(dev).to(dt)
    fn = lambda: ops.gemm_tn(dy, x512, conv=1)
elif which in ("dgrad1x1_n256", "dgrad1x1_nomask"):
    # decomposition of the data gradient's FETCH_SIZE: one N-tile per M-tile (the A panel is needed once) / no mask operand
    dh3 = torch.randn((M // 8, 1024), generator=g).to(dt).to(dev).repeat(8, 1)
    n = 256 if which.endswith("n256") else 512
    w3t = (torch.randn((n, 1024), generator=g) * 0.04).to(dev).to(dt)
    out = torch.empty((M, n), dtype=dt, device=dev)
    fn = lambda: ops.gemm_nt(dh3, w3t, None, out=out)
elif which == "dgrad1x1":
    # the centre head's masked 1024 -> 512 data gradient (two N-tiles per M-tile, K = 1024)
    dh3 = torch.randn((M // 8, 1024), generator=g).to(dt).to(dev).repeat(8, 1)
    w3t = (torch.randn((512, 1024), generator=g) * 0.04).to(dev).to(dt)
    out = torch.empty((M, 512), dtype=dt, device=dev)
    fn = lambda: ops.gemm_nt(dh3, w3t, None, aux=x512.view(M, 512), mask_relu=True, out=out)
else:
    w1 = (torch.randn((1024, 512), generator=g) * 0.04).to(dev).to(dt)
    out2 = torch.empty((M, 1024), dtype=dt, device=dev)
    fn = lambda: ops.gemm_nt(x512.view(M, 512), w1, torch.zeros(1024, device=dev), act=L.ACT_RELU, out=out2)
for _ in range(3):
    fn()
torch.cuda.synchronize()
