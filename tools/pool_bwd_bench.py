"""tools/pool_bwd_bench.py -- max-pool backward (first-max byte map + gather) on the training step's shapes (GPU only, tuning aid).
    python tools/pool_bwd_bench.py [--libs NAME,...]     experiment builds tools/libstep_amd_NAME.so beside the product library"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from step_amd import _capi, _lib, ops  # noqa: E402

# (name, N, D, H, W, C, k, s): the pools of one C4 step at 8 AVA clips per GPU
SHAPES = [("2a", 8, 18, 200, 200, 64, (1, 3, 3), (1, 2, 2)), ("3a", 8, 18, 100, 100, 192, (1, 3, 3), (1, 2, 2)), ("3b", 8, 18, 50, 50, 192, (3, 3, 3), (1, 1, 1)),
          ("3c", 8, 18, 50, 50, 256, (3, 3, 3), (1, 1, 1)), ("4a", 8, 18, 50, 50, 480, (3, 3, 3), (2, 2, 2)), ("4b", 8, 9, 25, 25, 480, (3, 3, 3), (1, 1, 1)),
          ("4f", 8, 9, 25, 25, 528, (3, 3, 3), (1, 1, 1)), ("5b@7x1080", 120, 9, 7, 7, 832, (3, 3, 3), (1, 1, 1))]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", default="")
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    libs = {"default": _lib.lib()}
    for nm in [n for n in a.libs.split(",") if n]:
        L_ = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libstep_amd_%s.so" % nm))
        _capi.declare(L_, strict=False)
        libs[nm] = L_
    print("%-12s %s" % ("pool", "  ".join("%22s" % n for n in libs)))
    for name, N, D, H, W, C, k, s in SHAPES:
        x = torch.relu(torch.randn(N, D, H, W, C, device="cuda")).bfloat16()
        y = ops.maxpool_tf(x, k, s)
        gy = torch.randn(y.shape, device="cuda").bfloat16()
        cells, ref = [], None
        for nm, L_ in libs.items():
            _lib._LIB = L_
            out = ops.maxpool_tf_backward(x, gy, k, s)
            torch.cuda.synchronize()
            if ref is None:
                ref = out
            else:
                assert torch.equal(out, ref), (name, nm)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                ops.maxpool_tf_backward(x, gy, k, s)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.iters
            gb = (x.numel() + 2 * gy.numel() + gy.numel() + x.numel()) * 2 / 1e9        # x, arg (written + read, 1 B), gy, gx
            cells.append("%9.3f ms %6.2f TB/s" % (ms, gb / ms))
        print("%-12s %s" % (name, "  ".join(cells)))
        _lib._LIB = libs["default"]


if __name__ == "__main__":
    main()
