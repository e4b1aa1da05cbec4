"""Per-kernel roofline table of the cfg2 train step from the committed rocprofv3 summary (profiles/rNN_pmc_traffic.json):
for the kernels whose shape is known from the workload (ViT-B/16, 384x384, batch 64: M = 9,437,184 head pixels, 36,928 tokens)
the algorithmic FLOPs and bytes, the achieved rates from the profiled average duration, the measured HBM-side traffic and MFMA
utilisation, and which roofline bounds the kernel.    python tools/roofline_table.py [profiles/r02_pmc_traffic.json]"""
import json
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "profiles/r06_pmc_traffic.json"
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 8      # steps in the profiled run (bench.py --steps 3 --warmup 2 + its 3 drained-queue steps)
ks = json.load(open(path))["kernels"]
M = 64 * 384 * 384
PEAK_TF, PEAK_TB = 2500.0, 8.0
# (kernel prefix, class, launches per step) -> (what, algorithmic FLOP, algorithmic bytes).  Round 6: the boundary-distance head runs
# collapsed in training, so the 3x3 conv / 512->1024 / 256->512 forward launches are the centre head's alone
ROWS = [
    (("gemm_nt256p_kernel<1, 3, 0", "large", 1), "head 3x3 conv 512->512, forward (centre head)", 2.0 * M * 512 * 4608, 2.0 * M * 512 * 2),
    (("gemm_nt256p_kernel<1, 3, 2", "large", 1), "head 3x3 conv, ReLU-masked data gradient (centre)", 2.0 * M * 512 * 4608, 3.0 * M * 512 * 2),
    (("gemm_tn256_kernel<1, true, 1", "large", 1), "head 3x3 conv, weight gradient (centre)", 2.0 * M * 512 * 4608, 2.0 * M * 512 * 2),
    (("gemm_nt256p_kernel<0, 3, 0, true", "large", 1), "1x1 512->1024 + fused output layer, h3 stored (centre)", 2.0 * M * 512 * 1024, M * 512 * 2 + M * 1024 * 2),
    (("gemm_nt256p_kernel<0, 3, 2", "large", 1), "1x1 1024->512 ReLU-masked data gradient (centre)", 2.0 * M * 512 * 1024, M * 1024 * 2 + 2.0 * M * 512 * 2),
    (("gemm_tn256_kernel<0, true, 0", "large", 1), "1x1 512->1024 weight gradient (centre)", 2.0 * M * 512 * 1024, M * 1024 * 2 + M * 512 * 2),
    (("head_out_bwd", "large", 1), "output layer backward: dh3, dW4 (centre)", 0.0, 2.0 * M * 1024 * 2),
    (("bilinear_fwd", "large", 1), "x2 resize of the centre head's first layer 192^2 -> 384^2, 512 ch, ReLU fused", 0.0, 1.25 * M * 512 * 2),
    (("bilinear_bwd", "large", 1), "its adjoint", 0.0, 1.25 * M * 512 * 2),
    (("attn_fwd", "large", 12), "attention forward, 64 x 12 heads x 577 tokens", 4.0 * 64 * 12 * 577 * 577 * 64, 4.0 * 36928 * 768 * 2),
    (("attn_bwd_dq", "large", 12), "attention backward dQ", 4.0 * 64 * 12 * 577 * 577 * 64, 6.0 * 36928 * 768 * 2),
    (("attn_bwd_dkv", "large", 12), "attention backward dK, dV", 6.0 * 64 * 12 * 577 * 577 * 64, 7.0 * 36928 * 768 * 2),
]


def find(key):
    pre, cls, n = key
    for k in ks:
        name = k["kernel"]
        if (name.startswith(pre) or pre in name) and k["class"] == cls and k["launches"] == n * STEPS:
            return k
    return None


print("| kernel (cfg2 step) | launches/step | avg ms | algorithmic TFLOP | achieved TFLOP/s (% of 2.5 PF) | MfmaUtil % | algorithmic GB | achieved TB/s on algorithmic bytes (% of 8) | measured HBM GB (FETCHx2 + WRITE) | bound |")
print("|---|---|---|---|---|---|---|---|---|---|")
for key, what, fl, by in ROWS:
    k = find(key)
    if k is None:
        continue
    t = k["avg_ms"] * 1e-3
    tf = fl / t / 1e12
    tb = by / t / 1e12
    meas = k.get("hbm_bytes_per_launch", 0.0) / 1e9
    fr_f, fr_b = tf / PEAK_TF, tb / PEAK_TB
    bound = "MFMA (power-capped, DESIGN 4)" if fr_f >= fr_b and fl > 0 else "HBM"
    if fl > 0 and abs(fr_f - fr_b) < 0.15 and fr_b > 0.3:
        bound = "MFMA and HBM within 15 %"
    if what.startswith("attention"):
        bound = "the tile loop's own dependent chain (ceiling table below)"
    print(f"| {what} | {k['launches'] / STEPS:.0f} | {k['avg_ms']:.2f} | {fl / 1e12:.2f} | {tf:.0f} ({100 * fr_f:.0f} %) | {k.get('mfma_util_percent', 0):.0f} | "
          f"{by / 1e9:.1f} | {tb:.2f} ({100 * fr_b:.0f} %) | {meas:.1f} | {bound} |")
