"""tools/gemm_probe.py -- what the vendor GEMM (torch.mm -> hipBLASLt / rocBLAS) reaches on the pointwise weight-gradient shapes,
beside step_conv_wgrad16 (diagnostic, GPU only): dW[Cout, Cin] = gy[M, Cout]^T x[M, Cin], K = M pixels."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from step_amd import ops  # noqa: E402

# (name, M, Cin, Cout)
SHAPES = [("5cf@7x1080", 1080 * 49, 832, 624), ("lc1@7x1080", 1080 * 49, 1088, 1024), ("lc_c1", 1080 * 49, 1024, 256), ("lc_c3", 1080 * 49, 256, 1024),
          ("4bf@400x8", 8 * 9 * 625, 480, 304), ("4b_b0", 8 * 9 * 625, 480, 192), ("3cf@400x8", 8 * 18 * 2500, 256, 288), ("3b_b1a", 8 * 18 * 2500, 192, 96),
          ("3bf (one read)", 8 * 18 * 2500, 192, 176), ("2b@400x8", 8 * 18 * 10000, 64, 64), ("4bf@400", 9 * 625, 480, 304)]


def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    print("%-16s %10s %10s %10s   (ms; TFLOP/s in brackets)" % ("shape", "step_amd", "torch.mm", "mm fp32out"))
    for name, M, ci, co in SHAPES:
        x = torch.randn(M, ci, device="cuda").bfloat16()
        gy = torch.randn(M, co, device="cuda").bfloat16()
        x5, g5 = x.view(1, 1, 1, M, ci), gy.view(1, 1, 1, M, co)
        gf = 2.0 * M * ci * co / 1e9
        t0 = timeit(lambda: ops.conv_wgrad16(x5, g5, co, (1, 1, 1)))
        t1 = timeit(lambda: torch.mm(gy.t(), x))
        out = torch.zeros(co, ci, device="cuda")
        try:
            t2 = timeit(lambda: torch.mm(gy.t(), x, out_dtype=torch.float32))
        except Exception:
            t2 = float("nan")
        a = ops.conv_wgrad16(x5, g5, co, (1, 1, 1)).view(co, ci)
        b = torch.mm(gy.t().float(), x.float())
        err = float((a - b).abs().max() / b.abs().max())
        print("%-16s %6.3f (%4.0f) %6.3f (%4.0f) %6.3f (%4.0f)   err %.1e" % (name, t0, gf / t0, t1, gf / t1, t2, gf / t2, err))


if __name__ == "__main__":
    main()
