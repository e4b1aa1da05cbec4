"""Is the head conv bound by cycles or by the power cap?  Times the 3x3 conv 512->512 (cfg2 shape) on random operands and on
all-zero operands (zeros draw far less power, so the chip holds its full clock: MI355X_MICROARCH.md 'DVFS give-back').
python tools/power_probe.py [B]   (A/B the kernel with UMR_NT256_STAGGER=0/1 in separate processes)
With the instrumented library (bash tools/probe/build_ts_lib.sh; UMR_LIB=unmore_amd/lib/libumr_ts.so) it also prints the clock the
chip holds inside the kernel: d(s_memtime) / d(s_memrealtime) x 100 MHz of workgroup 0 (guide: DVFS give-back, item 6)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from unmore_amd import ops, _lib as L
from tools.kbench import timeit

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
H = W = 384
dev = torch.device("cuda:0")
M = B * H * W
fl = 2.0 * M * 512 * 4608
g = torch.Generator().manual_seed(0)
out = torch.empty((M, 512), dtype=torch.bfloat16, device=dev)
bias = torch.zeros(512, device=dev)
for name, scale in (("random", 1.0), ("zeros", 0.0), ("random", 1.0)):
    x = (torch.randn((B, H, W, 512), generator=g) * scale).to(dev).to(torch.bfloat16)
    w = (torch.randn((512, 4608), generator=g) * 0.02 * scale).to(dev).to(torch.bfloat16)
    for _ in range(60):   # ~2 s of back-to-back launches: let the clock settle under this load
        ops.gemm_nt(x, w, bias, conv=1, out=out)
    t = timeit(lambda: ops.gemm_nt(x, w, bias, conv=1, out=out), n=9, warm=3)
    clk = ""
    if os.environ.get("UMR_LIB"):
        stamps = torch.zeros(144, dtype=torch.int64, device=dev)
        ops.gemm_nt(x, w, bias, conv=1, out=out, _stamps=stamps)
        torch.cuda.synchronize()
        st = stamps.cpu()
        clk = f"  in-kernel clock {float(st[138] - st[136]) / float(st[139] - st[137]) * 0.1:.3f} GHz"
    print(f"stagger={os.environ.get('UMR_NT256_STAGGER', '1')} conv3x3 512->512 NT on {name:6s}: {t:7.3f} ms {fl / t / 1e9:7.1f} TFLOP/s{clk}", flush=True)
    del x, w
