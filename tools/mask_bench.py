import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
from unmore_amd import ops, _lib as L
from kbench import timeit
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
B, H, W = 64, 384, 384
M = B * H * W
x = torch.randn(B // 8, H, W, 512, generator=g).to(dev).bfloat16().repeat(8, 1, 1, 1)
w3 = (torch.randn(512, 4608, generator=g) * 0.02).to(dev).bfloat16()
bias = torch.zeros(512, device=dev)
aux = x.view(M, 512)
out = torch.empty(M, 512, device=dev, dtype=torch.bfloat16)
fl = 2.0 * M * 512 * 4608
for name, fn in (("conv plain (bias+relu)", lambda: ops.gemm_nt(x, w3, bias, conv=1, act=L.ACT_RELU, out=out)),
                 ("conv relu-masked (dgrad form)", lambda: ops.gemm_nt(x, w3, None, conv=1, aux=aux, mask_relu=True, out=out))):
    t = timeit(fn, n=6)
    print(f"{name:34s} {t:7.3f} ms {fl / t / 1e9:7.1f} TF")
