import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unmore_amd import ops
torch.manual_seed(0)
A = torch.randn(4096, 768, device="cuda"); B = torch.randn(512, 768, device="cuda") * 0.05
ref = (A.double() @ B.double().t())
out = ops.gemm_nt(A, B, None)
scale = ref.abs().max().item()
print("UMR_F32_X3 =", os.environ.get("UMR_F32_X3", "1"), "max abs err / max|ref| = %.3e" % ((out.double() - ref).abs().max().item() / scale),
      "rms rel = %.3e" % ((out.double() - ref).norm() / ref.norm()).item())
t32 = (A @ B.t())
print("torch fp32 (rocBLAS): max abs err / max|ref| = %.3e rms rel = %.3e" % ((t32.double() - ref).abs().max().item() / scale, ((t32.double() - ref).norm() / ref.norm()).item()))
