"""tools/wgrad_bench.py -- timing of step_conv_wgrad on backbone layer shapes (diagnostic, GPU only)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from step_amd import ops  # noqa: E402

# (name, N, Cin, Cout, k, D, H, W)
LAYERS = [("2c@400", 1, 64, 192, 3, 18, 100, 100), ("3c_b1b@400", 1, 128, 192, 3, 18, 50, 50), ("4f_b1b@400", 1, 160, 320, 3, 9, 25, 25),
          ("3cf@400", 1, 256, 288, 1, 18, 50, 50), ("2c@224x8", 8, 64, 192, 3, 16, 56, 56), ("4f_b1b@224x8", 8, 160, 320, 3, 8, 14, 14)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    tdt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[a.dtype]
    for name, N, ci, co, k, D, H, W in LAYERS:
        x = torch.randn(N, D, H, W, ci, device="cuda").to(tdt)
        gy = torch.randn(N, D, H, W, co, device="cuda")
        ops.conv_wgrad(x, gy, co, (k, k, k))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            ops.conv_wgrad(x, gy, co, (k, k, k))
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        gf = 2.0 * N * D * H * W * ci * co * k ** 3 / 1e9
        print("%-14s %8.3f ms  %7.1f TFLOP/s" % (name, ms, gf / ms))


if __name__ == "__main__":
    main()
