"""Throughput of the existence classifier at the reference's batch shape (128 crops of 128x128, object_reasoning.py:492):
python tools/clf_bench.py [batch]   (MI355X)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from unmore_amd.binary_classifier import Binary_Classifier


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = Binary_Classifier(device="cuda:0", image_size=128, args=None).to(dev).eval()
    for m in net.modules():   # BatchNorm statistics of a trained checkpoint are not available: keep activations O(1)
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_var.fill_(1.0)
            m.weight.data.fill_(0.5)
    x = torch.rand(B, 3, 128, 128, device=dev)
    for dt in (torch.bfloat16, torch.float32):
        net.set_compute_dtype(dt)
        with torch.no_grad():
            for _ in range(3):
                net(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 10
            for _ in range(n):
                y = net(x)
            torch.cuda.synchronize()
        dtm = (time.perf_counter() - t0) / n
        # ResNet-50 forward: 4.1 GFLOP at 224^2 -> scales with pixels
        gflop = 2 * 4.09 * (128 * 128) / (224 * 224)
        print(f"Binary_Classifier {dt}: {1e3 * dtm:.2f} ms per {B}-crop batch = {B / dtm:.0f} crops/s ({gflop * B / dtm / 1e3:.1f} TFLOP/s)", flush=True)


if __name__ == "__main__":
    main()
