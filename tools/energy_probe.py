"""Energy per FLOP of the dominant kernels (under the board power cap joules/FLOP, not cycles, is the
resource the head conv runs out of -- MI355X_MICROARCH.md 'DVFS give-back', cdna_hip_programming.md 5.4 rule 28).

Runs ONE kernel back to back on random operands for --seconds while a thread samples the board power from the amdgpu hwmon
(power1_average, uW; fallback: rocm-smi --showpower), and reports ms / launch, TFLOP/s, average W, J / launch and pJ / FLOP.
Variants of the library are separate builds selected with UMR_LIB (tools/probe/build_exp_lib.sh) or run-time switches given
as environment variables, one process per variant on the SAME box (tools/probe/energy_ab.sh):

    python tools/energy_probe.py --kernel conv_nt|conv_nt_masked|conv_tn|g1x1_nt|conv_x3 [--seconds 6] [--tag name]"""
import argparse
import glob
import json
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from unmore_amd import ops, _lib as L


def power_reader():
    files = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average")) or sorted(glob.glob("/sys/class/hwmon/hwmon*/power1_average"))
    for f in files:
        try:
            float(open(f).read())
            return (lambda f=f: float(open(f).read()) * 1e-6), f
        except (OSError, ValueError):
            continue

    def smi():
        out = subprocess.run(["rocm-smi", "--showpower", "--json"], capture_output=True, text=True).stdout
        d = json.loads(out)
        for card in d.values():
            for k, v in card.items():
                if "ower" in k:
                    return float(v)
        raise RuntimeError("no power reading")
    return smi, "rocm-smi --showpower"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="conv_nt")
    ap.add_argument("--seconds", type=float, default=6.0)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--tag", default="")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    B, H, W = a.batch, 384, 384
    M = B * H * W
    g = torch.Generator().manual_seed(0)
    rnd = lambda *s, scale=1.0, dt=torch.bfloat16: (torch.randn(s, generator=g) * scale).to(dev).to(dt)
    if a.kernel in ("conv_nt", "conv_nt_masked", "conv_tn"):
        x = rnd(B, H, W, 512)
        w = rnd(512, 4608, scale=0.02)
        bias = torch.zeros(512, device=dev)
        out = torch.empty((M, 512), dtype=torch.bfloat16, device=dev)
        flop = 2.0 * M * 512 * 4608
        if a.kernel == "conv_nt":
            fn = lambda: ops.gemm_nt(x, w, bias, conv=1, act=L.ACT_RELU, out=out)
        elif a.kernel == "conv_nt_masked":
            aux = rnd(M, 512)
            fn = lambda: ops.gemm_nt(x, w, None, conv=1, aux=aux, mask_relu=True, out=out)
        else:
            dy = rnd(M, 512)
            fn = lambda: ops.gemm_tn(dy, x, conv=1)
    elif a.kernel == "g1x1_nt":
        x = rnd(M, 512)
        w = rnd(1024, 512, scale=0.04)
        bias = torch.zeros(1024, device=dev)
        out = torch.empty((M, 1024), dtype=torch.bfloat16, device=dev)
        flop = 2.0 * M * 512 * 1024
        fn = lambda: ops.gemm_nt(x, w, bias, act=L.ACT_RELU, out=out)
    elif a.kernel == "conv_x3":
        B = min(B, 16)
        M = B * H * W
        xp = ops.split3(rnd(B, H, W, 512, dt=torch.float32))
        wp = ops.split3(rnd(512, 4608, scale=0.02, dt=torch.float32))
        bias = torch.zeros(512, device=dev)
        flop = 2.0 * M * 512 * 4608
        fn = lambda: ops.gemm_nt_x3(xp, wp, bias, act=L.ACT_RELU, conv=1, out_planes=True)
    else:
        raise SystemExit("unknown kernel")
    read, src = power_reader()
    for _ in range(20):
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
    # settle ~1.5 s under load, then measure
    t_end = time.time() + 1.5
    while time.time() < t_end:
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    n, t0 = 0, time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.perf_counter() - t0 < a.seconds:
        for _ in range(10):
            fn()
        n += 10
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    stop.set()
    th.join()
    ms = e0.elapsed_time(e1) / n
    watts = sum(samples) / max(len(samples), 1)
    joule = watts * ms * 1e-3
    print(json.dumps({"tag": a.tag or os.environ.get("UMR_LIB", "default"), "kernel": a.kernel, "ms_per_launch": round(ms, 3),
                      "tflops": round(flop / ms / 1e9, 1), "avg_watts": round(watts, 1), "joule_per_launch": round(joule, 2),
                      "pj_per_flop": round(joule / flop * 1e12, 4), "power_samples": len(samples), "power_source": src, "launches": n}), flush=True)


if __name__ == "__main__":
    main()
