"""tools/pw_bench.py -- the pointwise convs of the training step (forward units and the data-gradient convs that accumulate through `res`)
alone, on the 8-clip AVA shapes (GPU only, tuning aid): kernel name, time, operand traffic over time."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from step_amd import _capi, _lib, ops  # noqa: E402

# (name, pixels as (N, D, H, W), Cin, Cout, accumulate through res)
SHAPES = [("3b b0 fwd", (8, 18, 50, 50), 192, 64, False), ("3b b1a fwd", (8, 18, 50, 50), 192, 96, False), ("3b b2a fwd", (8, 18, 50, 50), 192, 16, False),
          ("3c b0 fwd", (8, 18, 50, 50), 256, 128, False), ("3c b3 fwd", (8, 18, 50, 50), 256, 64, False),
          ("3c b0 dgrad +res", (8, 18, 50, 50), 128, 256, True), ("3c b2a dgrad +res", (8, 18, 50, 50), 32, 256, True), ("3c b3 dgrad", (8, 18, 50, 50), 64, 256, False),
          ("4b b0 fwd", (8, 9, 25, 25), 480, 192, False), ("4b b0 dgrad +res", (8, 9, 25, 25), 192, 480, True), ("4f b1a dgrad +res", (8, 9, 25, 25), 160, 528, True),
          ("2b fwd", (8, 18, 100, 100), 64, 64, False), ("2b dgrad", (8, 18, 100, 100), 64, 64, False), ("5c b0 @7x1080", (1080, 1, 7, 7), 832, 384, False)]


def main():
    L = _lib.lib()
    for name, (N, D, H, W), ci, co, acc in SHAPES:
        x = torch.randn(N, D, H, W, ci, device="cuda").bfloat16()
        w = ops.pack_conv_weight(torch.randn(co, ci, 1, 1, 1, device="cuda") * 0.05, torch.bfloat16)
        out = torch.randn(N, D, H, W, co, device="cuda").bfloat16()
        sc = torch.rand(co, device="cuda") + 0.5
        fn = (lambda: ops.conv_forward(x, w, co, (1, 1, 1), None, None, False, out, out)) if acc else (lambda: ops.conv_forward(x, w, co, (1, 1, 1), sc, sc, True, None, out))
        d = _capi.ConvDesc(dtype=_capi.BF16, N=N, D=D, H=H, W=W, Cin=ci, Cout=co, kd=1, kh=1, kw=1, x_cstride=ci, x_coff=0, y_cstride=co, y_coff=0,
                           res_cstride=co if acc else 0, res_coff=0, relu=0 if acc else 1, split=0, y2_cstride=0, y2_coff=0)
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
        gb = pix * (ci + co * (2 if acc else 1)) * 2 / 1e9
        print("%-20s %7.3f ms  %5.2f TB/s  %6.1f TFLOP/s  %s" % (name, ms, gb / ms, 2.0 * pix * ci * co / ms / 1e9, buf.value.decode()[11:60]))


if __name__ == "__main__":
    main()
