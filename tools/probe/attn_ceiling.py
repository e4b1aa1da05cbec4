"""Ceiling of the attention forward's tile loop at head dim 64 (round-5 review, item 7): the PRODUCT kernel's own loop
(unmore_amd/csrc/attention.hip::attn_fwd_bf16_kernel<2>: 32 queries per wave, 64-key tiles) rebuilt with -DUMR_ATTN_BARE, which
fills both LDS ring buffers once and then runs the loop with NO global or LDS-DMA traffic: QK^T MFMAs, the max3 chain and two lane
swaps, exp2, bf16 conversion, V^T transposing LDS reads, PV + row-sum MFMAs, one barrier per tile -- at 4, 2 and 1 workgroups per CU
(= waves per SIMD), beside the product build at the same shapes.

    python tools/probe/attn_ceiling.py            (builds unmore_amd/lib/libumr_exp.so; one child process per variant)

Per variant: ms per launch, algorithmic TFLOP/s (4 N^2 64 per head) and the share of the dense bf16 peak the matrix pipe spends on
EXECUTED MFMAs (36 per wave and tile for 32 algorithmic ones: the row sums ride on the matrix pipe) = what rocprofv3's MfmaUtil shows
for the kernel, up to the clock the chip holds."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r"""
import os, sys, json
sys.path.insert(0, %r)
sys.path.insert(0, os.path.join(%r, "tools"))
import torch
from unmore_amd import ops
from kbench import timeit
out = []
for (B, N, H) in ((8, 4096, 12), (64, 577, 12), (16, 1370, 16)):
    D = H * 64
    g = torch.Generator(device="cpu").manual_seed(0)
    qkv = (torch.randn(B * N, 3 * D, generator=g) * 0.5).cuda().bfloat16()
    ops.attention_fwd(qkv, B, N, H)
    t = min(timeit(lambda: ops.attention_fwd(qkv, B, N, H), n=30) for _ in range(3))
    fl = 4.0 * B * H * N * N * 64
    out.append({"B": B, "N": N, "heads": H, "ms": round(t, 4), "algorithmic_tflops": round(fl / t / 1e9, 1),
                "executed_mfma_share_of_2500": round(fl * 36 / 32 / t / 1e9 / 2500.0, 3)})
print(json.dumps(out))
""" % (ROOT, ROOT)


def run(tag, env):
    r = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
    if r.returncode != 0:
        print(json.dumps({"variant": tag, "error": r.stderr[-800:]}), flush=True)
        return
    for row in json.loads(r.stdout.strip().splitlines()[-1]):
        print(json.dumps({"variant": tag, **row}), flush=True)


if __name__ == "__main__":
    subprocess.run(["bash", os.path.join(ROOT, "tools", "probe", "build_exp_lib.sh"), "attention.hip", "-DUMR_ATTN_BARE"], check=True)
    exp = os.path.join(ROOT, "unmore_amd", "lib", "libumr_exp.so")
    run("product kernel (global K/V by LDS-DMA, 4 workgroups per CU)", {})
    run("bare loop, 4 workgroups per CU (4 waves per SIMD)", {"UMR_LIB": exp})
    run("bare loop, 2 workgroups per CU (2 waves per SIMD)", {"UMR_LIB": exp, "UMR_ATTN_BARE_LDS": "24576"})
    run("bare loop, 1 workgroup per CU (1 wave per SIMD)", {"UMR_LIB": exp, "UMR_ATTN_BARE_LDS": "98304"})
