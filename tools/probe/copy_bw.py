"""What a plain device copy reaches on this box, beside the streaming kernels' TB/s: python tools/probe/copy_bw.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tools.kbench import timeit
dev = torch.device("cuda:0")
for gb in (1, 4, 19.3):
    n = int(gb * 1e9 / 2)
    a = torch.empty(n, dtype=torch.bfloat16, device=dev).normal_()
    b = torch.empty_like(a)
    t = timeit(lambda: b.copy_(a))
    print(f"copy {gb} GB: {t:.3f} ms  {2 * n * 2 / t / 1e9:.2f} TB/s (read+write)")
    t = timeit(lambda: a.sum())
    print(f"read-only sum {gb} GB: {t:.3f} ms  {n * 2 / t / 1e9:.2f} TB/s")
    t = timeit(lambda: b.zero_())
    print(f"write-only fill {gb} GB: {t:.3f} ms  {n * 2 / t / 1e9:.2f} TB/s")
    del a, b
