"""The transformer MLP's two GELU-class GEMMs on the persistent 256x256 kernel: fc1 + GELU with the pre-activation saved, and the GELU'-masked
data gradient of fc2.  ViT-B at cfg2 (64 x 577 tokens) and ViT-L at cfg4's token count (1370 per image)."""
import sys
import torch
sys.path.insert(0, ".")
from unmore_amd import ops, _lib as L

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for name, M, D in (("ViT-B, 64 x 577 tokens", 64 * 577, 768), ("ViT-L, 16 x 1370 tokens", 16 * 1370, 1024)):
    x = torch.randn(M, D, generator=g).to(dev).bfloat16()
    w1 = (torch.randn(4 * D, D, generator=g) * 0.03).to(dev).bfloat16()
    b1 = torch.zeros(4 * D, device=dev)
    w2t = (torch.randn(4 * D, D, generator=g) * 0.03).to(dev).bfloat16()       # fc2 weight [D, 4D] transposed: dhp = dx . W2
    hpre = torch.randn(M, 4 * D, generator=g).to(dev).bfloat16()
    for label, fn in (("fc1 + GELU, pre-activation saved", lambda: ops.gemm_nt(x, w1, b1, act=L.ACT_GELU, c2_mode=2)),
                      ("fc1 + GELU", lambda: ops.gemm_nt(x, w1, b1, act=L.ACT_GELU)),
                      ("fc2 data gradient x GELU'", lambda: ops.gemm_nt(x, w2t, None, aux=hpre, mask_dgelu=True))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(30):
            fn()
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 30
        print(f"{name:26s} {label:34s} {ms * 1e3:8.1f} us  {2.0 * M * D * 4 * D / ms / 1e9:7.1f} TFLOP/s")
