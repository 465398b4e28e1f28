"""tools/tap_bench.py -- the 3x3x3 convs of the training step (forward units and their data gradients) alone, on the 8-clip AVA shapes and,
for comparison, on the 224 x 224 maps of C2 (GPU only, tuning aid)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from step_amd import _capi, _lib, ops  # noqa: E402

# (name, (N, D, H, W), Cin, Cout)
SHAPES = [("2c fwd @100", (8, 18, 100, 100), 64, 192), ("2c dgrad @100", (8, 18, 100, 100), 192, 64), ("3b_b1b fwd @50", (8, 18, 50, 50), 96, 128),
          ("3c_b1b fwd @50", (8, 18, 50, 50), 128, 192), ("3c_b1b dgrad @50", (8, 18, 50, 50), 192, 128), ("3c_b2b fwd @50", (8, 18, 50, 50), 32, 96),
          ("4b_b1b fwd @25", (8, 9, 25, 25), 96, 208), ("4f_b1b fwd @25", (8, 9, 25, 25), 160, 320), ("4f_b1b dgrad @25", (8, 9, 25, 25), 320, 160),
          ("5c_b1b fwd @13", (8, 5, 13, 13), 192, 384), ("5c_b1b fwd @7x1080", (120, 9, 7, 7), 192, 384),
          ("2c fwd @56 (C2)", (8, 16, 56, 56), 64, 192), ("3c_b1b fwd @28 (C2)", (8, 16, 28, 28), 128, 192), ("4f_b1b fwd @14 (C2)", (8, 8, 14, 14), 160, 320)]


def main():
    L = _lib.lib()
    for name, (N, D, H, W), ci, co in SHAPES:
        x = torch.randn(N, D, H, W, ci, device="cuda").bfloat16()
        w = ops.pack_conv_weight(torch.randn(co, ci, 3, 3, 3, device="cuda") * 0.02, torch.bfloat16)
        sc = torch.rand(co, device="cuda") + 0.5
        out = torch.empty(N, D, H, W, co, device="cuda", dtype=torch.bfloat16)
        fn = lambda: ops.conv_forward(x, w, co, (3, 3, 3), sc, sc, True, None, out)
        d = _capi.ConvDesc(dtype=_capi.BF16, N=N, D=D, H=H, W=W, Cin=ci, Cout=co, kd=3, kh=3, kw=3, x_cstride=ci, x_coff=0, y_cstride=co, y_coff=0,
                           res_cstride=0, res_coff=0, relu=1, split=0, y2_cstride=0, y2_coff=0)
        buf = ctypes.create_string_buffer(256)
        L.step_conv_kernel_name(ctypes.byref(d), buf, 256)
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        pix = N * D * H * W
        print("%-22s %7.3f ms  %7.1f TFLOP/s  %s" % (name, ms, 2.0 * pix * ci * co * 27 / ms / 1e9, buf.value.decode()[11:75]))


if __name__ == "__main__":
    main()
