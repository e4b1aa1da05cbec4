import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from unmore_amd import ops
    B, H, W = 32, 384, 384
    dev = torch.device("cuda:0"); dt = torch.bfloat16; M = B * H * W
    g = torch.Generator(device="cpu").manual_seed(0)
    x = torch.randn((B, H, W, 512), generator=g).to(dev).to(dt)
    dy = torch.randn((M, 512), generator=g).to(dev).to(dt)
    def t(fn, it=5):
        for _ in range(2): fn()
        torch.cuda.synchronize()
        ev = []
        for _ in range(it):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); ev.append((a, b))
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) for a, b in ev)
        return ts[len(ts) // 2], ts[0]
    med, mn = t(lambda: ops.gemm_tn(dy, x, conv=1))
    fl = 2.0 * M * 512 * 4608
    print(f"map={os.environ.get('UMR_TN_MAP','0')}: conv TN med {fl/med/1e9:7.1f} best {fl/mn/1e9:7.1f} TF", flush=True)
else:
    for rnd in range(2):
        for v in (sys.argv[1:] or ["0", "1", "2", "3"]):
            subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, UMR_TN_MAP=v))
