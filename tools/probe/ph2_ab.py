"""Two-phase against four-phase K-tile of the persistent 256x256 kernel on plain GEMMs (UMR_NT256_PH2=1 / 0: ops.set_debug_option; unset =
the host's choice: two-phase from 12 K-tile steps on), one process, same box.   python tools/probe/ph2_ab.py   (MI355X)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from unmore_amd import ops, _lib as L
from tools.kbench import timeit


def main():
    M = 64 * 384 * 384
    dev = torch.device("cuda:0")
    dt = torch.bfloat16
    g = torch.Generator(device="cpu").manual_seed(0)

    def rnd(*shape, scale=1.0, dtype=dt):
        return (torch.randn(shape, generator=g) * scale).to(dtype).to(dev)

    h2 = rnd(M // 8, 512).repeat(8, 1)
    dh3 = rnd(M // 8, 1024).repeat(8, 1)
    f256 = rnd(M // 8, 256).repeat(8, 1)
    w3, w3t, w1 = rnd(1024, 512, scale=0.04), rnd(512, 1024, scale=0.04), rnd(512, 256, scale=0.05)
    b3, b1 = torch.zeros(1024, device=dev), torch.zeros(512, device=dev)
    w4 = (torch.randn((2, 1024), generator=g) * 0.03).to(dev)
    out512 = torch.empty((M, 512), dtype=dt, device=dev)
    out1024 = torch.empty((M, 1024), dtype=dt, device=dev)
    cases = [
        ("head 1x1 256->512 (K-tiles 4)", lambda: ops.gemm_nt(f256, w1, b1, act=L.ACT_RELU, out=out512)),
        ("head 1x1 512->1024 + fused output, stored (8)", lambda: ops.gemm_nt(h2, w3, b3, act=L.ACT_RELU, red_w=w4, out=out1024)),
        ("head 1x1 512->1024 + fused output, no_store (8)", lambda: ops.gemm_nt(h2, w3, b3, act=L.ACT_RELU, red_w=w4, no_store=True)),
        ("head masked dgrad 1024->512 (16)", lambda: ops.gemm_nt(dh3, w3t, None, aux=h2, mask_relu=True, out=out512)),
    ]
    for name, Mt, D in (("ViT-B 36928 tokens", 64 * 577, 768), ("ViT-L/14 21920 tokens", 16 * 1370, 1024)):
        x, h = rnd(Mt, D), rnd(Mt, 4 * D)
        wq, wp, wf1, wf2 = rnd(3 * D, D, scale=0.03), rnd(D, D, scale=0.03), rnd(4 * D, D, scale=0.03), rnd(D, 4 * D, scale=0.02)
        bq, bf1 = torch.zeros(3 * D, device=dev), torch.zeros(4 * D, device=dev)
        res = rnd(Mt, D)
        cases += [
            (f"{name}: qkv ({D // 64})", lambda x=x, wq=wq, bq=bq: ops.gemm_nt(x, wq, bq)),
            (f"{name}: proj + residual ({D // 64})", lambda x=x, wp=wp, res=res: ops.gemm_nt(x, wp, None, aux=res)),
            (f"{name}: fc1 + GELU, saved pre-activation ({D // 64})", lambda x=x, wf1=wf1, bf1=bf1: ops.gemm_nt(x, wf1, bf1, act=L.ACT_GELU, c2_mode=2)),
            (f"{name}: fc2 ({4 * D // 64})", lambda h=h, wf2=wf2: ops.gemm_nt(h, wf2, None)),
            (f"{name}: fc1 data gradient, GELU' mask ({4 * D // 64})", lambda h=h, wf2=wf2, x=x: ops.gemm_nt(h, wf2, None)),
        ]
    # fp32-grade plane GEMMs of the reference recipe (1300 tokens, ViT-L) and of a head layer
    xs = ops.split3(rnd(1300, 1024, dtype=torch.float32))
    hs = ops.split3(rnd(1300, 4096, dtype=torch.float32))
    wqx, wf2x = ops.split3(rnd(3072, 1024, scale=0.03, dtype=torch.float32)), ops.split3(rnd(1024, 4096, scale=0.02, dtype=torch.float32))
    hp = ops.split3(rnd(20 * 128 * 128, 512, dtype=torch.float32))
    w3x = ops.split3(rnd(1024, 512, scale=0.04, dtype=torch.float32))
    cases += [
        ("fp32 planes: ViT-L qkv 1300 tokens (16 x 6, K-split)", lambda: ops.gemm_nt_x3(xs, wqx, None)),
        ("fp32 planes: ViT-L fc2 1300 tokens (64 x 6, K-split)", lambda: ops.gemm_nt_x3(hs, wf2x, None)),
        ("fp32 planes: head 512->1024 at 20 x 128^2 (8 x 6)", lambda: ops.gemm_nt_x3(hp, w3x, b3, act=L.ACT_RELU)),
    ]
    print(f"{'':62s} {'four-phase':>11s} {'two-phase':>11s} {'host pick':>11s}")
    for name, fn in cases:
        ts = []
        for v in ("0", "1", None):
            if v is None:
                ops.set_debug_option("UMR_NT256_PH2", None)
            else:
                ops.set_debug_option("UMR_NT256_PH2", v)
            ts.append(timeit(fn, n=9, warm=3))
        print(f"{name:62s} {ts[0] * 1e3:9.1f} us {ts[1] * 1e3:9.1f} us {ts[2] * 1e3:9.1f} us   two/four {ts[1] / ts[0]:.3f}", flush=True)
    ops.set_debug_option("UMR_NT256_PH2", None)


if __name__ == "__main__":
    main()
