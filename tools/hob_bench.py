"""head_out_bwd at the cfg2 centre-head shape (M = B*384*384 rows of 1024 bf16 channels): python tools/hob_bench.py [B]
Times the K = 1024 kernel and, with UMR_HEAD_OUT_BWD_GENERIC set, the generic one; checks them against each other."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from unmore_amd import ops, _lib as L
from tools.kbench import timeit

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
H = W = 384
K = 1024
dev = torch.device("cuda:0")
torch.manual_seed(0)
M = B * H * W
h = torch.randn(M, K, device=dev).relu_().bfloat16()   # post-ReLU activations, as in the step
w = torch.randn(2, K, device=dev) * 0.05
dout = torch.randn(B, 2, H, W, device=dev)
yout = torch.tanh(torch.randn(B, 2, H, W, device=dev))
res = {}
for name, env in (("k1024", None), ("generic", "1")):
    if env:
        ops.set_debug_option("UMR_HEAD_OUT_BWD_GENERIC", env)
    else:
        ops.set_debug_option("UMR_HEAD_OUT_BWD_GENERIC", None)
    dw = torch.zeros(2, K, device=dev)
    db = torch.zeros(2, device=dev)
    dh = ops.head_out_bwd(h, w, dout, yout, L.ACT_TANH, True, dw, db)
    res[name] = (dh.clone(), dw.clone(), db.clone())
    del dh
    t = timeit(lambda: ops.head_out_bwd(h, w, dout, yout, L.ACT_TANH, True, dw, db))
    print(f"{name}: {t:.3f} ms  {2 * M * K * 2 / t / 1e9:.2f} TB/s")
a, b = res["k1024"], res["generic"]
print("dh equal:", bool((a[0] == b[0]).all()), " max|ddw|/max|dw|:", float((a[1] - b[1]).abs().max() / b[1].abs().max()),
      " ddb:", float((a[2] - b[2]).abs().max() / b[2].abs().max()))
