"""tools/stem_wgrad_bench.py -- step_stem_wgrad (fp32 MFMA, gradient re-read per filter row) against step_stem_wgrad16
(16-bit MFMA, operands read once) on one clip shape (GPU only, tuning aid).

    python tools/stem_wgrad_bench.py [--n 1 --t 36 --hw 400]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from step_amd import ops  # noqa: E402


def timeit(fn, iters=10):
    fn(); fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1)
    ap.add_argument("--t", type=int, default=36)
    ap.add_argument("--hw", type=int, default=400)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    for dt in (torch.bfloat16, torch.float16):
        x = torch.randn(a.n, a.t, 3, a.hw, a.hw, device=dev).to(dt)
        To, Ho = (a.t - 2) // 2 + 1, (a.hw - 2) // 2 + 1
        g = (torch.randn(a.n, To, Ho, Ho, 64, device=dev) * 0.1).to(dt)
        g32 = g.float()
        ref = ops.stem_wgrad(x, g32, 64)
        got = ops.stem_wgrad16(x, g, 64)
        err = float((got - ref).norm() / ref.norm())
        t0 = timeit(lambda: ops.stem_wgrad(x, g32, 64))
        t1 = timeit(lambda: ops.stem_wgrad16(x, g, 64))
        gf = 2.0 * a.n * To * Ho * Ho * 64 * 1029 / 1e9
        print("%s  [%d,%d,3,%d,%d]: stem_wgrad %.1f us (%.0f TF)   stem_wgrad16 %.1f us (%.0f TF)   rel diff %.2e" % (
            str(dt)[6:], a.n, a.t, a.hw, a.hw, t0, gf / t0 * 1e3, t1, gf / t1 * 1e3, err))


if __name__ == "__main__":
    main()
