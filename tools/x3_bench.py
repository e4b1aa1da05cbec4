"""Same-process A/B of the fp32-grade GEMM forms at the shapes of the reference recipe's fp32 step (dpt_large, 20 x 128^2: 1300
tokens; README.md:148-155): bf16-plane operands on the 256x256 kernels (K-split for the small problems) against the 128x128
kernels that split f32 operands in registers.  Prints microseconds per call and f32-equivalent TFLOP/s (peak 2500 / 6 = 417)."""
import sys

import torch

sys.path.insert(0, ".")
from unmore_amd import ops  # noqa: E402
from unmore_amd import _lib as L  # noqa: E402


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(0)
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
    rows = []
    # plain NT GEMMs of the transformer (M tokens) and of the heads (M pixels)
    for name, M, N, K in (("qkv", 1300, 3072, 1024), ("proj", 1300, 1024, 1024), ("fc1", 1300, 4096, 1024), ("fc2", 1300, 1024, 4096),
                          ("vit-b qkv 3250", 3250, 2304, 768), ("vit-b fc2 3250", 3250, 768, 3072),
                          ("head 256->512", 327680, 512, 256), ("head 512->1024", 327680, 1024, 512)):
        A, B, bias = rnd(M, K), rnd(N, K) * K ** -0.5, rnd(N)
        Ap, Bp = ops.split3(A), ops.split3(B)
        t_pl = timeit(lambda: ops.gemm_nt_x3(Ap, Bp, bias))
        t_128 = timeit(lambda: ops.gemm_nt(A, B, bias))
        fl = 2.0 * M * N * K
        rows.append((f"NT {name} {M}x{N}x{K}", t_pl, t_128, fl))
    # 3x3 convs: DPT maps and the head conv
    for name, nb, H, W, Cin, N in (("rcu 8x8", 20, 8, 8, 256, 256), ("rcu 16x16", 20, 16, 16, 256, 256), ("rcu 32x32", 20, 32, 32, 256, 256),
                                   ("rcu 64x64", 20, 64, 64, 256, 256), ("head conv", 20, 128, 128, 512, 512)):
        x, w, bias = rnd(nb, H, W, Cin), rnd(N, 9 * Cin) * (9 * Cin) ** -0.5, rnd(N)
        xp, wp = ops.split3(x), ops.split3(w)
        t_pl = timeit(lambda: ops.gemm_nt_x3(xp, wp, bias, conv=1, act=L.ACT_RELU), reps=10)
        t_128 = timeit(lambda: ops.gemm_nt(x, w, bias, conv=1, act=L.ACT_RELU), reps=5)
        rows.append((f"conv {name} {nb}x{H}x{W}x{Cin}->{N}", t_pl, t_128, 2.0 * nb * H * W * N * 9 * Cin))
    # weight gradients
    for name, M, N, K in (("qkv", 1300, 3072, 1024), ("proj", 1300, 1024, 1024), ("fc1", 1300, 4096, 1024), ("fc2", 1300, 1024, 4096),
                          ("head W3", 327680, 1024, 512), ("head W1", 327680, 512, 256)):
        dY, X = rnd(M, N), rnd(M, K)
        dYp, Xp = ops.split3(dY), ops.split3(X)
        db = torch.empty(N, device=dev)
        t_pl = timeit(lambda: ops.gemm_tn(dYp, Xp, dbias=db, x3=True), reps=10)
        t_128 = timeit(lambda: ops.gemm_tn(dY, X, dbias=db), reps=5)
        rows.append((f"TN {name} {M}x{N}x{K}", t_pl, t_128, 2.0 * M * N * K))
    for name, nb, H, W, Cin, N in (("rcu 16x16", 20, 16, 16, 256, 256), ("rcu 64x64", 20, 64, 64, 256, 256), ("head conv", 20, 128, 128, 512, 512)):
        x, dy = rnd(nb, H, W, Cin), rnd(nb * H * W, N)
        xp, dyp = ops.split3(x), ops.split3(dy)
        t_pl = timeit(lambda: ops.gemm_tn(dyp, xp, conv=1, x3=True), reps=5)
        t_128 = timeit(lambda: ops.gemm_tn(dy, x, conv=1), reps=3)
        rows.append((f"TN conv {name} {nb}x{H}x{W}x{Cin}->{N}", t_pl, t_128, 2.0 * nb * H * W * N * 9 * Cin))
    print(f"{'problem':46s} {'planes us':>10s} {'TF/s':>7s} {'128^2 f32 us':>13s} {'TF/s':>7s}  speed-up")
    for n, a, b, fl in rows:
        print(f"{n:46s} {a:10.1f} {fl / a / 1e6:7.1f} {b:13.1f} {fl / b / 1e6:7.1f}  {b / a:5.2f}x")


if __name__ == "__main__":
    main()
