"""Energy-table rows for the MFMA shape and BK: builds tools/probe/mfma_shape.hip on the GPU box and runs each
variant back to back for --seconds on random bf16 operands while the board power is sampled (tools/energy_probe.power_reader).
One JSON line per variant: ms per launch, TFLOP/s, W, pJ/FLOP.   python tools/probe/mfma_shape.py [--seconds 3]"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
from energy_probe import power_reader  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=3.0)
a = ap.parse_args()
so = "/tmp/libmfma_shape.so"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o", so,
                os.path.join(ROOT, "tools", "probe", "mfma_shape.hip")], check=True)
lib = ctypes.CDLL(so)
lib.mfma_shape_run.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda:0")
init = torch.randn(4096 * 8, dtype=torch.float32, device=dev).to(torch.bfloat16)      # 4096 x 16 B of random bf16
blocks, ktiles = 512, 20000                                                          # two workgroups per CU, ~30 ms per launch
out = torch.empty(blocks * 512, dtype=torch.float32, device=dev)
flop = 2.0 * blocks * 256 * 256 * 64 * ktiles
read, src = power_reader()
names = {0: "mfma 16x16x32, BK 64 (the kernel's form)", 1: "mfma 32x32x16, BK 64", 2: "mfma 16x16x32, BK 32 (two barriers per 64 of K)"}
for rnd in range(2):
    for v in (0, 1, 2):
        fn = lambda: lib.mfma_shape_run(v, init.data_ptr(), out.data_ptr(), blocks, ktiles, torch.cuda.current_stream().cuda_stream)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        samples, stop = [], threading.Event()

        def sampler():
            while not stop.is_set():
                try:
                    samples.append(read())
                except Exception:
                    pass
                time.sleep(0.05)
        t_end = time.time() + 1.0
        while time.time() < t_end:
            fn()
            torch.cuda.synchronize()
        th = threading.Thread(target=sampler, daemon=True)
        th.start()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n, t0 = 0, time.perf_counter()
        e0.record()
        while time.perf_counter() - t0 < a.seconds:
            for _ in range(5):
                fn()
            n += 5
            torch.cuda.synchronize()
        e1.record()
        torch.cuda.synchronize()
        stop.set()
        th.join()
        ms = e0.elapsed_time(e1) / n
        watts = sum(samples) / max(len(samples), 1)
        print(json.dumps({"tag": names[v], "round": rnd, "kernel": "bare loop, 128x64 per wave, fragments from LDS, no global traffic",
                          "ms_per_launch": round(ms, 3), "tflops": round(flop / ms / 1e9, 1), "avg_watts": round(watts, 1),
                          "pj_per_flop": round(watts * ms * 1e-3 / flop * 1e12, 4), "power_source": src, "launches": n}), flush=True)
