"""Per-tile overhead of the 256x256 NT kernel: 1x1 head GEMMs at cfg2 size and a K sweep (run on the MI355X).
python tools/kbench4.py [B]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from unmore_amd import ops, _lib as L
from kbench import timeit


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    dev = torch.device("cuda:0")
    dt = torch.bfloat16
    M = B * 384 * 384
    g = torch.Generator(device="cpu").manual_seed(0)

    def rnd(*shape, scale=1.0):
        return (torch.randn(shape, generator=g) * scale).to(dev).to(dt)

    a = rnd(M // 64, 2048).repeat(64, 1)
    for K, N in ((64, 512), (128, 512), (256, 512), (512, 512), (1024, 512), (2048, 512), (512, 1024), (512, 256)):
        A = a[:, :K].contiguous()
        w = rnd(N, K, scale=0.03)
        bias = torch.zeros(N, device=dev)
        out = torch.empty((M, N), dtype=dt, device=dev)
        t = timeit(lambda: ops.gemm_nt(A, w, bias, act=L.ACT_RELU, out=out))
        tiles = ((M + 255) // 256) * ((N + 255) // 256)
        per_tile_us = t * 1e3 / (tiles / 256.0)
        gb = (M * K + M * N) * 2 / 1e9
        print(f"K={K:5d} N={N:5d}: {t:8.3f} ms {2.0 * M * K * N / t / 1e9:8.1f} TFLOP/s  {gb / t:6.2f} TB/s  {per_tile_us:6.2f} us/tile/CU ({K // 64} k-tiles)", flush=True)
        del A, out


if __name__ == "__main__":
    main()
