"""tools/pool_bench.py -- the max pools of the C2 backbone (and the AVA-shaped maps with --set c3) one by one: us per launch and
algorithmic GB/s (input + output bytes).  GPU only, tuning aid."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from step_amd import ops  # noqa: E402

# (name, C, D, H, W, kernel, stride) per clip
C2 = [("pool2a", 64, 16, 112, 112, (1, 3, 3), (1, 2, 2)), ("pool3a", 192, 16, 56, 56, (1, 3, 3), (1, 2, 2)),
      ("3b_pool", 192, 16, 28, 28, (3, 3, 3), (1, 1, 1)), ("3c_pool", 256, 16, 28, 28, (3, 3, 3), (1, 1, 1)),
      ("pool4a", 480, 16, 28, 28, (3, 3, 3), (2, 2, 2)),
      ("4b_pool", 480, 8, 14, 14, (3, 3, 3), (1, 1, 1)), ("4c_pool", 512, 8, 14, 14, (3, 3, 3), (1, 1, 1)),
      ("4f_pool", 528, 8, 14, 14, (3, 3, 3), (1, 1, 1))]
C3 = [("pool2a", 64, 18, 200, 200, (1, 3, 3), (1, 2, 2)), ("pool3a", 192, 18, 100, 100, (1, 3, 3), (1, 2, 2)),
      ("3b_pool", 192, 18, 50, 50, (3, 3, 3), (1, 1, 1)), ("3c_pool", 256, 18, 50, 50, (3, 3, 3), (1, 1, 1)),
      ("pool4a", 480, 18, 50, 50, (3, 3, 3), (2, 2, 2)), ("4b_pool", 480, 9, 25, 25, (3, 3, 3), (1, 1, 1)),
      ("4f_pool", 528, 9, 25, 25, (3, 3, 3), (1, 1, 1))]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--set", default="c2")
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    tot = 0.0
    for name, C, D, H, W, k, s in (C2 if a.set == "c2" else C3):
        x = torch.randn(a.batch, D, H, W, C, device=dev).to(torch.bfloat16)
        y = ops.maxpool_tf(x, k, s)
        for _ in range(3):
            ops.maxpool_tf(x, k, s, y)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            ops.maxpool_tf(x, k, s, y)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / a.iters * 1e3
        tot += us
        print("%-8s C=%3d %2dx%3dx%3d k%s s%s  %7.1f us  %6.0f GB/s" % (name, C, D, H, W, "".join(map(str, k)), "".join(map(str, s)), us,
                                                                    (x.numel() + y.numel()) * 2 / us * 1e-3))
    print("total %.1f us" % tot)


if __name__ == "__main__":
    main()
