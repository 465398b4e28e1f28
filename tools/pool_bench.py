"""tools/pool_bench.py -- per-layer timing of the TF-SAME max pools of the backbone at the C2 geometry through the
C ABI (diagnostic, GPU only).  Prints microseconds and effective GB/s (algorithmic bytes: read once + write once)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from step_amd import ops  # noqa: E402

# (name, C, D, H, W, k, s) per clip at T=32, 224x224
POOLS = [
    ("pool1", 64, 16, 112, 112, (1, 3, 3), (1, 2, 2)), ("pool2", 192, 16, 56, 56, (1, 3, 3), (1, 2, 2)),
    ("pool3", 480, 16, 28, 28, (3, 3, 3), (2, 2, 2)),
    ("3b_p", 192, 16, 28, 28, (3, 3, 3), (1, 1, 1)), ("3c_p", 256, 16, 28, 28, (3, 3, 3), (1, 1, 1)),
    ("4b_p", 480, 8, 14, 14, (3, 3, 3), (1, 1, 1)), ("4c_p", 512, 8, 14, 14, (3, 3, 3), (1, 1, 1)),
    ("4f_p", 528, 8, 14, 14, (3, 3, 3), (1, 1, 1)),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    tot = 0.0
    for name, C, D, H, W, k, s in POOLS:
        x = torch.randn(a.batch, D, H, W, C, device="cuda").to(torch.bfloat16)
        y = ops.maxpool_tf(x, k, s)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            ops.maxpool_tf(x, k, s, out=y)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / a.iters * 1e3
        gb = (x.numel() + y.numel()) * 2 / 1e9
        tot += us
        print("%-6s C=%3d %2dx%3dx%3d  %7.1f us  %7.1f GB/s" % (name, C, D, H, W, us, gb / (us * 1e-6)))
    print("total %.1f us" % tot)


if __name__ == "__main__":
    main()
