"""Per-step kernel time of a profiled bench.py run by group, from profiles/rNN_pmc_traffic*.json (tools/profile_round.sh):
python tools/step_breakdown.py profiles/r05_pmc_traffic.json [steps_in_the_run=8]"""
import json
import sys

d = json.load(open(sys.argv[1]))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rows = d["kernels"]


def grp(k):
    n, big = k["kernel"], k.get("class") == "large"
    if n.startswith("gemm_nt256p_kernel<1, 3") and big and k["avg_ms"] > 5:
        return "head 3x3 conv NT (centre head: forward, masked data gradient; + the boundary-distance head's forward in factored mode)"
    if n.startswith("gemm_tn256_kernel<1") and k["avg_ms"] > 5:
        return "head 3x3 conv TN (weight gradient)"
    if n.startswith("gemm_nt256p_kernel<0, 3, 0, true") or (n.startswith("gemm_nt256p_kernel<0, 3, 2") and k["avg_ms"] > 5) or \
            (n.startswith("gemm_tn256_kernel<0") and k["avg_ms"] > 5) or n.startswith("head_out"):
        return "centre-head 1x1 group (512->1024 + fused output, masked dgrad, W3 wgrad, head_out_bwd)"
    if n.startswith("attn"):
        return "attention"
    if n.startswith("bilinear"):
        return "resizes"
    if n.startswith("ln_") or "ln_bwd" in n or "ln_fwd" in n:
        return "LayerNorm"
    if "adam" in n or "loss" in n:
        return "loss + Adam"
    if n.startswith("tn_reduce") or "reduce" in n:
        return "slab reductions"
    if n.startswith("permute4") or "cast" in n or "pixel_shuffle" in n or "segsum" in n or "patchify" in n:
        return "packing / weight refresh / small element-wise"
    if n.startswith("gemm"):
        return "other GEMMs (transformer, DPT, 256->512 layers on the small map)"
    return "other"


acc = {}
for k in rows:
    g = grp(k)
    acc[g] = acc.get(g, 0.0) + k["total_ms"]
tot = sum(acc.values())
print(f"{sys.argv[1]}: kernel ms per step over {steps} steps (top-40 kernels of the run, {d['command']})")
for g, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"  {v / steps:8.2f} ms  {100 * v / tot:5.1f} %  {g}")
print(f"  {tot / steps:8.2f} ms  total of the listed kernels")
print()
print(f"{'kernel':62s} {'class':5s} {'n/step':>6s} {'avg ms':>8s} {'GB fetched':>10s} {'GB written':>10s} {'MfmaUtil':>8s}")
for k in rows[:26]:
    print(f"{k['kernel'][:62]:62s} {k.get('class', '')[:5]:5s} {k['launches'] / steps:6.1f} {k['avg_ms']:8.3f} {k.get('fetch_bytes_per_launch', 0) / 1e9:10.2f} "
          f"{k.get('write_bytes_per_launch', 0) / 1e9:10.2f} {k.get('mfma_util_percent', 0):8.1f}")
