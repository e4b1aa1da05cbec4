import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
from unmore_amd import ops, _lib as L
from kbench import timeit
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
M, N, K = 36928, 3072, 768
A = torch.randn(M, K, generator=g).to(dev).bfloat16()
B = (torch.randn(N, K, generator=g) * 0.03).to(dev).bfloat16()
bias = torch.zeros(N, device=dev)
aux = torch.randn(M, N, generator=g).to(dev).bfloat16()
fl = 2.0 * M * N * K
for name, fn in (("bias only (fast class)", lambda: ops.gemm_nt(A, B, bias)),
                 ("bias + GELU, pre-act copy (fc1 fwd)", lambda: ops.gemm_nt(A, B, bias, act=L.ACT_GELU, c2_mode=2)),
                 ("bias + GELU", lambda: ops.gemm_nt(A, B, bias, act=L.ACT_GELU))):
    t = timeit(fn, n=20)
    print(f"{name:40s} {t:7.3f} ms {fl / t / 1e9:7.1f} TF")
A2 = torch.randn(M, N, generator=g).to(dev).bfloat16(); B2 = (torch.randn(K, N, generator=g) * 0.02).to(dev).bfloat16()
t = timeit(lambda: ops.gemm_nt(A2, B2, None, aux=aux[:, :K].contiguous(), mask_dgelu=False), n=20)
print(f"{'fc2-shaped dgrad add-aux':40s} {t:7.3f} ms {fl / t / 1e9:7.1f} TF")
a3 = torch.randn(M, N, generator=g).to(dev).bfloat16()
A3 = torch.randn(M, K, generator=g).to(dev).bfloat16(); B3 = (torch.randn(N, K, generator=g) * 0.03).to(dev).bfloat16()
t = timeit(lambda: ops.gemm_nt(A3, B3, None, aux=a3, mask_dgelu=True), n=20)
print(f"{'dgelu-mask epilogue (fc1 pre-act grad)':40s} {t:7.3f} ms {fl / t / 1e9:7.1f} TF")
