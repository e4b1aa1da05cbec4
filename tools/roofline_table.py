"""Per-kernel roofline table of the cfg2 train step from the committed rocprofv3 summary (profiles/rNN_pmc_traffic.json):
for the kernels whose shape is known from the workload (ViT-B/16, 384x384, batch 64: M = 9,437,184 head pixels, 36,928 tokens)
the algorithmic FLOPs and bytes, the achieved rates from the profiled average duration, the measured HBM-side traffic and MFMA
utilisation, and which roofline bounds the kernel.    python tools/roofline_table.py [profiles/r02_pmc_traffic.json]"""
import json
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "profiles/r02_pmc_traffic.json"
ks = json.load(open(path))["kernels"]
M = 64 * 384 * 384
PEAK_TF, PEAK_TB = 2500.0, 8.0
# (kernel prefix, class, launches in the 5 profiled steps) -> (what, algorithmic FLOP, algorithmic bytes)
ROWS = [
    (("gemm_nt256p_kernel<1, 3, 0", "large", 10), "head 3x3 conv 512->512, forward (both heads)", 2.0 * M * 512 * 4608, 2.0 * M * 512 * 2),
    (("gemm_nt256p_kernel<1, 3, 2", "large", 5), "head 3x3 conv, ReLU-masked data gradient (centre)", 2.0 * M * 512 * 4608, 3.0 * M * 512 * 2),
    (("gemm_tn256_kernel<1, true, true", "large", 5), "head 3x3 conv, weight gradient (centre)", 2.0 * M * 512 * 4608, 2.0 * M * 512 * 2),
    (("gemm_nt256p_kernel<0, 3, 0, true", "large", 10), "1x1 512->1024 + fused output layer (centre stores h3, sdf does not)", 2.0 * M * 512 * 1024, M * 512 * 2 + 0.5 * M * 1024 * 2),
    (("gemm_nt256p_kernel<0, 3, 2", "large", 5), "1x1 1024->512 ReLU-masked data gradient (centre)", 2.0 * M * 512 * 1024, M * 1024 * 2 + 2.0 * M * 512 * 2),
    (("gemm_tn256_kernel<0, true, false", "large", 5), "1x1 512->1024 weight gradient (centre)", 2.0 * M * 512 * 1024, M * 1024 * 2 + M * 512 * 2),
    (("head_out_bwd", "large", 5), "output layer backward: dh3, dW4 (centre)", 0.0, 2.0 * M * 1024 * 2),
    (("gemm_nt256p_kernel<0, 3, 0, false", "large", 10), "1x1 256->512 forward (both heads)", 2.0 * M * 256 * 512, M * 256 * 2 + M * 512 * 2),
    (("gemm_nt256p_kernel<0, 3, 0, false", "large", 5), "feature-map gradient 512->256 (centre)", 2.0 * M * 256 * 512, M * 256 * 2 + M * 512 * 2),
    (("gemm_tn256_kernel<0, true, false", "small", 5), "1x1 256->512 weight gradient (centre)", 2.0 * M * 256 * 512, M * 256 * 2 + M * 512 * 2),
    (("lh_bwd_data", "large", 5), "boundary-distance head, algebraic backward: feature-map gradient (accumulate)", 2.0 * M * 256 * 9, 2.0 * M * 256 * 2),
    (("lh_bwd_weight", "large", 5), "boundary-distance head, algebraic backward: pixel reductions", 2.0 * M * 256 * 9, M * 256 * 2),
    (("bilinear_fwd", "large", 5), "final x2 upsample 192^2 -> 384^2, 256 ch (largest of 5 resizes)", 0.0, 1.25 * M * 256 * 2),
    (("bilinear_bwd", "large", 5), "its adjoint", 0.0, 1.25 * M * 256 * 2),
]


def find(key):
    pre, cls, n = key
    for k in ks:
        name = k["kernel"]
        if (name.startswith(pre) or pre in name) and k["class"] == cls and k["launches"] == n:
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
    print(f"| {what} | {k['launches'] / 5:.0f} | {k['avg_ms']:.2f} | {fl / 1e12:.2f} | {tf:.0f} ({100 * fr_f:.0f} %) | {k.get('mfma_util_percent', 0):.0f} | "
          f"{by / 1e9:.1f} | {tb:.2f} ({100 * fr_b:.0f} %) | {meas:.1f} | {bound} |")
