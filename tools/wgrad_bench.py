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
          ("2c@400x8", 8, 64, 192, 3, 18, 100, 100), ("3c_b1b@400x8", 8, 128, 192, 3, 18, 50, 50), ("3b_b1b@400x8", 8, 96, 128, 3, 18, 50, 50),
          ("4f_b1b@400x8", 8, 160, 320, 3, 9, 25, 25), ("4c_b1b@400x8", 8, 112, 224, 3, 9, 25, 25), ("5c_b1b@7x1080", 1080, 192, 384, 3, 1, 7, 7), ("lc3x3@7x1080", 1080, 256, 256, (1, 3, 3), 1, 7, 7),
          ("4bf@400x8", 8, 480, 304, 1, 9, 25, 25), ("3b_b0@400x8pw", 8, 192, 64, 1, 18, 50, 50), ("3b_b1a@400x8pw", 8, 192, 96, 1, 18, 50, 50), ("3c_b0@400x8pw", 8, 256, 128, 1, 18, 50, 50),
          ("3c_b2a@400x8pw", 8, 256, 32, 1, 18, 50, 50), ("4b_b0@400x8pw", 8, 480, 192, 1, 9, 25, 25), ("4b_b2a@400x8pw", 8, 480, 16, 1, 9, 25, 25), ("2b@400x8pw", 8, 64, 64, 1, 18, 100, 100),
          ("lc1@7x1080pw", 1080, 1088, 1024, 1, 1, 7, 7), ("lc_c1@7x1080pw", 1080, 1024, 256, 1, 1, 7, 7), ("lc_c3@7x1080pw", 1080, 256, 1024, 1, 1, 7, 7), ("5cf@7x1080", 1080, 832, 624, 1, 1, 7, 7),
          ("3b_b2b@400", 1, 16, 32, 3, 18, 50, 50), ("4bf@400", 1, 480, 304, 1, 9, 25, 25), ("2b@400", 1, 64, 64, 1, 18, 100, 100), ("4b_b1b@400", 1, 96, 208, 3, 9, 25, 25), ("5b_b1b@13", 1, 160, 320, 3, 9, 13, 13)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--only", default="", help="comma-separated substrings of layer names")
    ap.add_argument("--lds-only", action="store_true")
    ap.add_argument("--graph", action="store_true", help="time the calls replayed from a captured HIP graph (no host time between launches)")
    ap.add_argument("--libs", default="", help="comma-separated experiment builds (tools/libstep_amd_NAME.so, `make EXP=NAME EXPFLAGS=...`) timed beside the product library")
    a = ap.parse_args()
    tdt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[a.dtype]
    # variants: fp32-dY kernel; with 16-bit storage also the 16-bit-MFMA forms (per tap / one row of taps per job)
    variants = [("fp32 mfma", None)]
    if tdt != torch.float32:
        variants += [("16-bit mfma, per tap", "0"), ("16-bit mfma, LDS tiles", "lds")]
    print("%-14s %s" % ("layer", "  ".join("%26s" % v[0] for v in variants)))
    if a.lds_only:
        variants = [v for v in variants if v[1] == "lds"]
    if a.libs:
        import ctypes
        base = _lib.lib()
        libs = {"default": base}
        for nm in a.libs.split(","):
            L_ = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libstep_amd_%s.so" % nm))
            _capi.declare(L_, strict=False)
            libs[nm] = L_
        variants = [("LDS tiles, lib=" + nm, "lib:" + nm) for nm in libs]
    for name, N, ci, co, k, D, H, W in LAYERS:
        if a.only and not any(o in name for o in a.only.split(",")):
            continue
        x = torch.randn(N, D, H, W, ci, device="cuda").to(tdt)
        gy = torch.randn(N, D, H, W, co, device="cuda")
        gy16 = gy.to(tdt)
        kk = (k, k, k) if isinstance(k, int) else tuple(k)
        gf = 2.0 * N * D * H * W * ci * co * kk[0] * kk[1] * kk[2] / 1e9
        cells, ref = [], None
        for _, row in variants:
            if row is None:
                fn = lambda: ops.conv_wgrad(x, gy16.float(), co, kk)     # (same rounded dY as the 16-bit forms)
                g32 = gy16.float()
                fn = lambda: ops.conv_wgrad(x, g32, co, kk)
            elif row.startswith("lib:"):
                _lib._LIB = libs[row[4:]]
                fn = lambda: ops.conv_wgrad16(x, gy16, co, kk)
            else:
                _capi.set_option(_lib.lib(), "wgrad16_lds", 1 if row == "lds" else 0)
                fn = lambda: ops.conv_wgrad16(x, gy16, co, kk)
            out = fn()
            torch.cuda.synchronize()
            if ref is None:
                ref = out
            else:
                err = float((out - ref).abs().max() / ref.abs().max())
                assert err < 1e-3, (name, row, err)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if a.graph:
                gr = torch.cuda.CUDAGraph()
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    fn()
                    with torch.cuda.graph(gr, stream=side):
                        for _ in range(a.iters):
                            fn()
                torch.cuda.current_stream().wait_stream(side)
                gr.replay()
                torch.cuda.synchronize()
                e0.record()
                gr.replay()
                e1.record()
            else:
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
