"""Small-M, long-K GEMMs of the reference recipe (1300 tokens, ViT-L): device time per launch with and without split-K.
python tools/probe/small_gemm.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unmore_amd import ops

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)


def bench(fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for dt in (torch.bfloat16, torch.float32):
    for (M, N, K) in [(1300, 1024, 4096), (1300, 1024, 3072), (1300, 1024, 1024), (1300, 4096, 1024), (1300, 3072, 1024), (130, 384, 1536)]:
        A = torch.randn((M, K), generator=g).to(dev).to(dt)
        B = (torch.randn((N, K), generator=g) * K ** -0.5).to(dev).to(dt)
        bias = torch.zeros(N, device=dev)
        out = torch.empty((M, N), device=dev, dtype=dt)
        res = []
        for env in ("0", None, "2", "3", "5", "8"):
            if env is None:
                ops.set_debug_option("UMR_NT_SPLITK", None)
            else:
                ops.set_debug_option("UMR_NT_SPLITK", env)
            res.append(f"{env or 'auto'}: {bench(lambda: ops.gemm_nt(A, B, bias, out=out)):6.1f}")
        ops.set_debug_option("UMR_NT_SPLITK", None)
        print(f"{str(dt)[6:]:9s} M={M} N={N} K={K}  us per launch  " + "  ".join(res), flush=True)


# 3x3 convs on small maps (the DPT fusion blocks at 224^2 batch 2 / 128^2 batch 20): [nb, H, W, Cin] -> N
for dt in (torch.bfloat16, torch.float32):
    for (nb, H, W, Cin, N) in [(2, 7, 7, 256, 256), (2, 14, 14, 256, 256), (2, 28, 28, 256, 256), (2, 56, 56, 256, 256), (20, 16, 16, 256, 256), (20, 32, 32, 256, 256)]:
        A = torch.randn((nb, H, W, Cin), generator=g).to(dev).to(dt)
        B = (torch.randn((N, 9 * Cin), generator=g) * (9 * Cin) ** -0.5).to(dev).to(dt)
        bias = torch.zeros(N, device=dev)
        res = []
        for env in ("0", None, "2", "3", "4", "6", "8"):
            if env is None:
                ops.set_debug_option("UMR_NT_SPLITK", None)
            else:
                ops.set_debug_option("UMR_NT_SPLITK", env)
            res.append(f"{env or 'auto'}: {bench(lambda: ops.gemm_nt(A, B, bias, conv=1)):6.1f}")
        ops.set_debug_option("UMR_NT_SPLITK", None)
        tiles = ((nb * H * W + 127) // 128) * ((N + 127) // 128)
        print(f"{str(dt)[6:]:9s} conv3x3 [{nb},{H},{W},{Cin}]->{N} ({tiles} tiles)  us per launch, UMR_NT_SPLITK =  " + "  ".join(res), flush=True)
