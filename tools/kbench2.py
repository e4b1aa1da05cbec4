"""A/B of kernel variants in ONE process (interleaved rounds): python tools/kbench2.py"""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# each variant needs its own process (the env var is read once); run them interleaved for fairness
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from unmore_amd import ops, _lib as L
    B, H, W = 16, 384, 384
    dev = torch.device("cuda:0"); dt = torch.bfloat16; M = B * H * W
    g = torch.Generator(device="cpu").manual_seed(0)
    x = torch.randn((B, H, W, 512), generator=g).to(dev).to(dt)
    w3 = (torch.randn((512, 4608), generator=g) * 0.02).to(dev).to(dt)
    out = torch.empty((M, 512), dtype=dt, device=dev)
    bias = torch.zeros(512, device=dev)
    n = 8192
    A = torch.randn((n, n), generator=g).to(dev).to(dt); Bm = torch.randn((n, n), generator=g).to(dev).to(dt)
    C = torch.empty((n, n), dtype=dt, device=dev)
    def t(fn, it=6):
        for _ in range(2): fn()
        torch.cuda.synchronize()
        ev = []
        for _ in range(it):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); ev.append((a, b))
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) for a, b in ev)
        return ts[len(ts) // 2], ts[0]
    c_med, c_min = t(lambda: ops.gemm_nt(x, w3, bias, conv=1, act=L.ACT_RELU, out=out))
    g_med, g_min = t(lambda: ops.gemm_nt(A, Bm, None, out=C))
    fl = 2.0 * M * 512 * 4608
    print(f"var={os.environ.get('UMR_NT256_VAR','0')}: conv med {fl/c_med/1e9:7.1f} best {fl/c_min/1e9:7.1f} TF | gemm8192 med {2.0*n**3/g_med/1e9:7.1f} best {2.0*n**3/g_min/1e9:7.1f} TF", flush=True)
else:
    variants = sys.argv[1:] or ["0", "1", "2", "3"]
    for rnd in range(2):
        for v in variants:
            env = dict(os.environ, UMR_NT256_VAR=v)
            subprocess.run([sys.executable, __file__, "child"], env=env)
