"""tools/act_grad_bench.py -- step_act_grad alone on the training step's shapes (GPU only, tuning aid): operand traffic over time."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from step_amd import ops  # noqa: E402

SHAPES = [("stem", 8 * 18 * 200 * 200, 64), ("2c", 8 * 18 * 100 * 100, 192), ("3c concat", 8 * 18 * 50 * 50, 480), ("3c bottlenecks", 8 * 18 * 50 * 50, 160),
          ("4f concat", 8 * 9 * 25 * 25, 832), ("5c@7x1080", 1080 * 49, 1024), ("3c concat, 1 clip", 18 * 50 * 50, 480)]


def main():
    for name, M, C in SHAPES:
        y = torch.relu(torch.randn(M, C, device="cuda")).bfloat16()
        gy = torch.randn(M, C, device="cuda").bfloat16()
        sc = torch.rand(C, device="cuda") + 0.5
        fn = lambda: ops.act_grad(y, gy, sc, True, want_f32=False, want_act=True)
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print("%-20s %9d x %4d  %7.3f ms  %5.2f TB/s" % (name, M, C, ms, M * C * 6 / ms / 1e9))


if __name__ == "__main__":
    main()
