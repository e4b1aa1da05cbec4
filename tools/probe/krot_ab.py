"""K rotation between the N-tiles of an M-tile in the persistent 256x256 GEMM (csrc/gemm_nt256p.hip, UMR_NT256_KROT, read per launch):
the workgroups that compute the N-tiles of one M-tile stream the same A panel side by side; started a few K-tiles apart, one of them
fetches a panel chunk from HBM and the others find it in L2.   python tools/probe/krot_ab.py [B]   (MI355X)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from unmore_amd import ops, _lib as L
from tools.kbench import timeit


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    M = B * 384 * 384
    dev = torch.device("cuda:0")
    dt = torch.bfloat16
    g = torch.Generator(device="cpu").manual_seed(0)

    def rnd(*shape, scale=1.0):
        return (torch.randn(shape, generator=g) * scale).to(dt).to(dev)

    h2 = rnd(M // 8, 512).repeat(8, 1)
    dh3 = rnd(M // 8, 1024).repeat(8, 1)
    w3 = rnd(1024, 512, scale=0.04)
    w3t = rnd(512, 1024, scale=0.04)
    b3 = torch.zeros(1024, device=dev)
    w4 = (torch.randn((2, 1024), generator=g) * 0.03).to(dev)
    out512 = torch.empty((M, 512), dtype=dt, device=dev)
    Mt = 64 * 577
    xt = rnd(Mt, 768)
    wfc1 = rnd(3072, 768, scale=0.03)
    bfc1 = torch.zeros(3072, device=dev)
    wqkv = rnd(2304, 768, scale=0.03)
    ht = rnd(Mt, 3072)
    wfc2 = rnd(768, 3072, scale=0.02)
    cases = [
        ("masked dgrad 1024->512 (M = %d)" % M, 2.0 * M * 512 * 1024, lambda: ops.gemm_nt(dh3, w3t, None, aux=h2, mask_relu=True, out=out512)),
        ("fused-output fwd 512->1024, no_store", 2.0 * M * 512 * 1024, lambda: ops.gemm_nt(h2, w3, b3, act=L.ACT_RELU, red_w=w4, no_store=True)),
        ("256..: ViT-B fc1 + GELU (36928 x 3072 x 768)", 2.0 * Mt * 3072 * 768, lambda: ops.gemm_nt(xt, wfc1, bfc1, act=L.ACT_GELU)),
        ("ViT-B qkv (36928 x 2304 x 768)", 2.0 * Mt * 2304 * 768, lambda: ops.gemm_nt(xt, wqkv, None)),
        ("ViT-B fc2 (36928 x 768 x 3072)", 2.0 * Mt * 768 * 3072, lambda: ops.gemm_nt(ht, wfc2, None)),
    ]
    ref = {}
    for rot in (0, 1, 2, 3, 4, 6):
        os.environ["UMR_NT256_KROT"] = str(rot)
        for name, fl, fn in cases:
            t = timeit(fn, n=7, warm=2)
            r = fn()
            r = next(t for t in r if t is not None) if isinstance(r, tuple) else r
            key = name
            if rot == 0:
                ref[key] = r.float().clone() if r.numel() < (1 << 26) else r[:1 << 16].float().clone()
                d = 0.0
            else:
                cur = r.float() if r.numel() < (1 << 26) else r[:1 << 16].float()
                d = (cur - ref[key]).abs().max().item()
            print(f"rot {rot}  {name:48s} {t:8.3f} ms  {fl / t / 1e9:8.1f} TFLOP/s   max|diff vs rot 0| {d:.3g}", flush=True)
    os.environ.pop("UMR_NT256_KROT", None)


if __name__ == "__main__":
    main()
