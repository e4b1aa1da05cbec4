"""Where a tile's time goes in the persistent 256x256 NT kernel: s_memtime stamps written by workgroup 0 / thread 0.
Needs a library built with -DUMR_NT256P_TIMESTAMPS:
    bash tools/probe/build_ts_lib.sh          (here or on the GPU box; writes unmore_amd/lib/libumr_ts.so)
    UMR_LIB=unmore_amd/lib/libumr_ts.so python tools/probe/ts_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from unmore_amd import ops, _lib as L

assert os.environ.get("UMR_LIB"), "run with UMR_LIB=<instrumented library>"
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
M = 64 * 384 * 384 // 4
SHAPES = ((768, 512, 0), (512, 1024, 0), (512, 1024, 3), (768, 512, 2), (768, 768, 1))
if len(sys.argv) > 1 and sys.argv[1] == "dgrad":     # the centre head's 1024 -> 512 data gradient at the full cfg2 row count, plain and masked
    SHAPES = ((1024, 512, 0), (1024, 512, 2), (1024, 256, 0))
    M *= 4
if len(sys.argv) > 1 and sys.argv[1] == "red":       # the 512 -> 1024 layer with the fused output layer: plain, stored, no_store (aux 4)
    SHAPES = ((512, 1024, 0), (512, 1024, 3), (512, 1024, 4))
for K, N, aux in SHAPES:   # aux 3 = fused row reduction
    A = torch.randn(M // 64, K, generator=g).to(dev).bfloat16().repeat(64, 1)
    w = (torch.randn(N, K, generator=g) * 0.03).to(dev).bfloat16()
    bias = torch.zeros(N, device=dev)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    ax = torch.randn(M // 64, N, generator=g).to(dev).bfloat16().repeat(64, 1) if aux in (1, 2) else None
    rw = torch.ones(2, N, device=dev) if aux in (3, 4) else None
    stamps = torch.zeros(16 * 8 + 8, dtype=torch.int64, device=dev)
    for _ in range(3):
        ops.gemm_nt(A, w, bias, act=(L.ACT_RELU if aux in (0, 3, 4) else L.ACT_NONE), out=(None if aux == 4 else out), aux=ax, mask_relu=(aux == 2),
                    red_w=rw, no_store=(aux == 4), _stamps=stamps)
    torch.cuda.synchronize()
    ph = stamps.cpu()[128:136]
    t = stamps.cpu()[:128].view(16, 8)
    print(f"K={K} N={N} aux mode {aux}: shader-clock cycles per tile segment (tiles 2..7 of workgroup 0)")
    if aux in (0, 3) and int(ph[7]) > 0:   # 4-phase K-tile (plain GEMM): after-sync -> after-MFMA-block per phase, and the gaps between
        d = [int(ph[i + 1] - ph[i]) for i in range(7)]
        print(f"  one K-tile (tile 4, k-tile 5): MFMA blocks {d[0]} {d[2]} {d[4]} {d[6]}   read+wait+barrier gaps {d[1]} {d[3]} {d[5]}")
    for it in range(2, 8):
        r, nxt = t[it], t[it + 1][0]
        print(f"  tile {it}: K loop {int(r[4] - r[0]):6d}  epilogue {int(r[6] - r[4]):6d}  sync+tail {int(nxt - r[6]):5d} | k-tile0 {int(r[2] - r[1]):5d}  k-tile1 {int(r[3] - r[2]):5d}  other k-tiles {int(r[4] - r[3]):6d}  "
              f"pass0 {int(r[5] - r[4]):5d}  passes1-3 {int(r[6] - r[5]):5d}  tail {int(nxt - r[6]):4d}   total {int(nxt - r[0]):6d}")
