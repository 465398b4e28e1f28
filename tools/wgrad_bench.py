"""tools/wgrad_bench.py -- timing of step_conv_wgrad on backbone layer shapes (diagnostic, GPU only)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from step_amd import _capi, _lib, ops  # noqa: E402

# (name, N, Cin, Cout, k, D, H, W)
LAYERS = [("2c@400", 1, 64, 192, 3, 18, 100, 100), ("3c_b1b@400", 1, 128, 192, 3, 18, 50, 50), ("4f_b1b@400", 1, 160, 320, 3, 9, 25, 25),
          ("3cf@400", 1, 256, 288, 1, 18, 50, 50), ("2c@224x8", 8, 64, 192, 3, 16, 56, 56), ("4f_b1b@224x8", 8, 160, 320, 3, 8, 14, 14),
          ("3b_b2b@400", 1, 16, 32, 3, 18, 50, 50), ("4bf@400", 1, 480, 304, 1, 9, 25, 25), ("2b@400", 1, 64, 64, 1, 18, 100, 100), ("4b_b1b@400", 1, 96, 208, 3, 9, 25, 25), ("5b_b1b@13", 1, 160, 320, 3, 9, 13, 13)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    tdt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[a.dtype]
    # variants: fp32-dY kernel; with 16-bit storage also the 16-bit-MFMA forms (per tap / one row of taps per job)
    variants = [("fp32 mfma", None)]
    if tdt != torch.float32:
        variants += [("16-bit mfma, per tap", "0"), ("16-bit mfma, LDS tiles", "lds")]
    print("%-14s %s" % ("layer", "  ".join("%26s" % v[0] for v in variants)))
    for name, N, ci, co, k, D, H, W in LAYERS:
        x = torch.randn(N, D, H, W, ci, device="cuda").to(tdt)
        gy = torch.randn(N, D, H, W, co, device="cuda")
        gy16 = gy.to(tdt)
        gf = 2.0 * N * D * H * W * ci * co * k ** 3 / 1e9
        cells, ref = [], None
        for _, row in variants:
            if row is None:
                fn = lambda: ops.conv_wgrad(x, gy16.float(), co, (k, k, k))     # (same rounded dY as the 16-bit forms)
                g32 = gy16.float()
                fn = lambda: ops.conv_wgrad(x, g32, co, (k, k, k))
            else:
                _capi.set_option(_lib.lib(), "wgrad16_lds", 1 if row == "lds" else 0)
                fn = lambda: ops.conv_wgrad16(x, gy16, co, (k, k, k))
            out = fn()
            torch.cuda.synchronize()
            if ref is None:
                ref = out
            else:
                err = float((out - ref).abs().max() / ref.abs().max())
                assert err < 1e-3, (name, row, err)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.iters
            cells.append("%10.3f ms %7.1f TFLOP/s" % (ms, gf / ms))
        print("%-14s %s" % (name, "  ".join(cells)))


if __name__ == "__main__":
    main()
