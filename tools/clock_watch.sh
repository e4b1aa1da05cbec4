#!/bin/bash
# sample GPU clocks / power while the conv kernel loops:  bash tools/clock_watch.sh   (GPU box, repo root)
python3 - <<'PY' &
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from unmore_amd import ops, _lib as L
B, H, W = 64, 384, 384
dev = torch.device("cuda:0")
x = torch.randn(B, H, W, 512, device=dev).bfloat16()
w = (torch.randn(512, 4608, device=dev) * 0.02).bfloat16()
b = torch.zeros(512, device=dev)
out = torch.empty(B * H * W, 512, device=dev, dtype=torch.bfloat16)
t0 = time.time()
n = 0
while time.time() - t0 < 25:
    for _ in range(10):
        ops.gemm_nt(x, w, b, conv=1, act=L.ACT_RELU, out=out)
    torch.cuda.synchronize()
    n += 10
print("conv launches", n, "avg ms", 1e3 * (time.time() - t0) / n, flush=True)
PY
PID=$!
sleep 12
for i in 1 2 3 4; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power|power" | head -8
  echo ---
  sleep 2
done
wait $PID
