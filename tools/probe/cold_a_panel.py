"""Is the K loop of a plain GEMM on the persistent 256x256 kernel bound by the latency of a COLD A panel?  The centre head's
1024 -> 512 data gradient shape (two N-tiles per M-tile) at row counts whose A operand is HBM-streamed (19.3 GB) or held by the
256-MB infinity cache, per tile and workgroup.   python tools/probe/cold_a_panel.py   (MI355X)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from unmore_amd import ops
from tools.kbench import timeit

dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
for K, N in ((1024, 512), (512, 1024)):
    w = (torch.randn((N, K), generator=g) * 0.04).to(torch.bfloat16).to(dev)
    for M in (9437184, 2359296, 524288, 262144, 131072, 65536):
        a = torch.randn((min(M, 65536), K), generator=g).to(torch.bfloat16).to(dev)
        A = a.repeat(M // a.shape[0], 1) if M > a.shape[0] else a
        out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        t = timeit(lambda: ops.gemm_nt(A, w, None, out=out), n=9, warm=3)
        tiles = (M // 256) * (N // 256)
        per_cu = tiles / 256.0
        print(f"K={K} N={N} M={M:8d}  A = {M * K * 2 / 2**20:8.0f} MiB  {t * 1e3:9.1f} us  {2.0 * M * N * K / t / 1e9:7.1f} TFLOP/s  "
              f"{tiles:6d} tiles = {per_cu:6.1f} per CU -> {t * 1e3 / max(per_cu, 1):6.2f} us per tile", flush=True)
        del A, out
