"""Micro-benchmarks of the dominant kernels at cfg2 shapes (run on the MI355X):
python tools/kbench.py [B]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from unmore_amd import ops, _lib as L


def timeit(fn, n=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    H = W = 384
    dev = torch.device("cuda:0")
    dt = torch.bfloat16
    M = B * H * W
    g = torch.Generator(device="cpu").manual_seed(0)

    def rnd(*shape, scale=1.0):
        return (torch.randn(shape, generator=g) * scale).to(dev).to(dt)

    x512 = rnd(B, H, W, 512)
    w3 = rnd(512, 9 * 512, scale=0.02)
    bias = torch.zeros(512, device=dev)
    out = torch.empty((M, 512), dtype=dt, device=dev)
    t = timeit(lambda: ops.gemm_nt(x512, w3, bias, conv=1, act=L.ACT_RELU, out=out))
    fl = 2.0 * M * 512 * 4608
    print(f"conv3x3 512->512 fwd  NT : {t:8.3f} ms  {fl / t / 1e9:8.1f} TFLOP/s")
    dy = rnd(M, 512)
    t = timeit(lambda: ops.gemm_tn(dy, x512, conv=1))
    print(f"conv3x3 512->512 wgrad TN: {t:8.3f} ms  {fl / t / 1e9:8.1f} TFLOP/s")
    w1 = rnd(1024, 512, scale=0.04)
    b1 = torch.zeros(1024, device=dev)
    out2 = torch.empty((M, 1024), dtype=dt, device=dev)
    a2 = x512.view(M, 512)
    t = timeit(lambda: ops.gemm_nt(a2, w1, b1, act=L.ACT_RELU, out=out2))
    fl2 = 2.0 * M * 512 * 1024
    print(f"1x1 512->1024 fwd     NT : {t:8.3f} ms  {fl2 / t / 1e9:8.1f} TFLOP/s")
    w1t = rnd(512, 1024, scale=0.04)
    t = timeit(lambda: ops.gemm_nt(out2, w1t, None, out=out))
    print(f"1x1 1024->512 dgrad   NT : {t:8.3f} ms  {fl2 / t / 1e9:8.1f} TFLOP/s")
    t = timeit(lambda: ops.gemm_tn(out2, a2))
    print(f"1x1 512->1024 wgrad   TN : {t:8.3f} ms  {fl2 / t / 1e9:8.1f} TFLOP/s")
    # plain square GEMM reference point
    n = 8192
    A = rnd(n, n)
    Bm = rnd(n, n)
    C = torch.empty((n, n), dtype=dt, device=dev)
    t = timeit(lambda: ops.gemm_nt(A, Bm, None, out=C))
    print(f"gemm 8192^3           NT : {t:8.3f} ms  {2.0 * n ** 3 / t / 1e9:8.1f} TFLOP/s")
    # ViT-B qkv at B=64
    Mt = 64 * 577
    xa = rnd(Mt, 768)
    wq = rnd(2304, 768, scale=0.03)
    oq = torch.empty((Mt, 2304), dtype=dt, device=dev)
    t = timeit(lambda: ops.gemm_nt(xa, wq, None, out=oq))
    print(f"qkv 36928x2304x768    NT : {t:8.3f} ms  {2.0 * Mt * 2304 * 768 / t / 1e9:8.1f} TFLOP/s")


if __name__ == "__main__":
    main()
