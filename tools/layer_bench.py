"""tools/layer_bench.py -- per-layer timing of the backbone kernels at the C2 geometry
(batch x [3,32,224,224]) through the C ABI.  Diagnostic tool (GPU only)."""
import argparse
import ctypes
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from step_amd import _capi, _lib  # noqa: E402

TORCH_DT = {_capi.F32: torch.float32, _capi.BF16: torch.bfloat16, _capi.F16: torch.float16}

# (name, Cin, Cout, k, D, H, W) per clip at T=32, 224x224
LAYERS = [
    ("2b_1x1", 64, 64, 1, 16, 56, 56), ("2c_3x3", 64, 192, 3, 16, 56, 56),
    ("3b_b0", 192, 64, 1, 16, 28, 28), ("3b_b1a", 192, 96, 1, 16, 28, 28), ("3b_b1b", 96, 128, 3, 16, 28, 28),
    ("3b_b2a", 192, 16, 1, 16, 28, 28), ("3b_b2b", 16, 32, 3, 16, 28, 28), ("3b_b3", 192, 32, 1, 16, 28, 28),
    ("3c_b0", 256, 128, 1, 16, 28, 28), ("3c_b1a", 256, 128, 1, 16, 28, 28), ("3c_b1b", 128, 192, 3, 16, 28, 28),
    ("3c_b2a", 256, 32, 1, 16, 28, 28), ("3c_b2b", 32, 96, 3, 16, 28, 28), ("3c_b3", 256, 64, 1, 16, 28, 28),
    ("4b_b0", 480, 192, 1, 8, 14, 14), ("4b_b1b", 96, 208, 3, 8, 14, 14), ("4b_b2b", 16, 48, 3, 8, 14, 14),
    ("4c_b1b", 112, 224, 3, 8, 14, 14), ("4d_b1b", 128, 256, 3, 8, 14, 14), ("4e_b1b", 144, 288, 3, 8, 14, 14),
    ("4f_b0", 528, 256, 1, 8, 14, 14), ("4f_b1b", 160, 320, 3, 8, 14, 14), ("4f_b2b", 32, 128, 3, 8, 14, 14),
]


def time_it(fn, iters):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--dtype", type=int, default=_capi.BF16)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--only", default="", help="comma list of layer names (skips stem/pool)")
    ap.add_argument("--custom", action="append", default=[], help="name,Cin,Cout,k,D,H,W (repeatable)")
    a = ap.parse_args()
    L = _lib.lib()
    dt, tdt = a.dtype, TORCH_DT[a.dtype]
    dev = torch.device("cuda:0")
    st = _lib.stream_ptr()
    B = a.batch
    tot_ms, tot_gf = 0.0, 0.0
    only = set(x for x in a.only.split(",") if x)
    if a.custom:
        cl = []
        for c in a.custom:
            f = c.split(",")
            cl.append((f[0],) + tuple(int(v) for v in f[1:]))
        run_layers(L, dt, tdt, dev, st, B, a, cl)
        return
    if only:
        run_layers(L, dt, tdt, dev, st, B, a, [l for l in LAYERS if l[0] in only])
        return
    # stem
    x = (torch.rand(B, 32, 3, 224, 224, device=dev) * 2 - 1).to(tdt)
    w = torch.randn(64, 3, 7, 7, 7, device=dev) * 0.03
    wp = torch.empty(L.step_stem_packed_elems(64), dtype=tdt, device=dev)
    _capi.check(L.step_stem_pack_weight(_lib.dptr(w), 64, dt, _lib.dptr(wp), st), "pack")
    sc = torch.ones(64, device=dev)
    sh = torch.zeros(64, device=dev)
    y = torch.empty(B, 16, 112, 112, 64, dtype=tdt, device=dev)
    ms = time_it(lambda: _capi.check(L.step_stem_forward(dt, _lib.dptr(x), B, 32, 224, 224, _lib.dptr(wp), _lib.dptr(sc), _lib.dptr(sh), 1, 64, _lib.dptr(y), 64, 0, st), "stem"), a.iters)
    gf = 2 * B * 16 * 112 * 112 * 64 * 1029 / 1e9
    print("%-8s %8.3f ms %8.1f TFLOP/s (useful)" % ("stem", ms, gf / ms))
    tot_ms += ms
    tot_gf += gf
    # pool 2a
    y2 = torch.empty(B, 16, 56, 56, 64, dtype=tdt, device=dev)
    ms = time_it(lambda: _capi.check(L.step_maxpool3d_tf(dt, _lib.dptr(y), B, 16, 112, 112, 64, 64, 0, 1, 3, 3, 1, 2, 2, _lib.dptr(y2), 64, 0, st), "pool"), a.iters)
    byts = (y.numel() + y2.numel()) * y.element_size()
    print("%-8s %8.3f ms %8.1f GB/s" % ("pool2a", ms, byts / ms / 1e6))
    tot_ms += ms
    run_layers(L, dt, tdt, dev, st, B, a, LAYERS, tot_ms, tot_gf)


def run_layers(L, dt, tdt, dev, st, B, a, layers, tot_ms=0.0, tot_gf=0.0):
    for name, ci, co, k, D, H, W in layers:
        x = torch.randn(B, D, H, W, ci, device=dev).to(tdt)
        w = torch.randn(co, ci, k, k, k, device=dev) * (1.0 / (ci * k ** 3) ** 0.5)
        wp = torch.empty(L.step_conv_packed_elems(co, ci, k, k, k), dtype=tdt, device=dev)
        _capi.check(L.step_conv_pack_weight(_lib.dptr(w), co, ci, k, k, k, dt, None, _lib.dptr(wp), st), "pack")
        sc = torch.ones(co, device=dev)
        sh = torch.zeros(co, device=dev)
        y = torch.empty(B, D, H, W, co, dtype=tdt, device=dev)
        d = _capi.ConvDesc(dtype=dt, N=B, D=D, H=H, W=W, Cin=ci, Cout=co, kd=k, kh=k, kw=k, x_cstride=ci, x_coff=0,
                           y_cstride=co, y_coff=0, res_cstride=0, res_coff=0, relu=1, split=0, y2_cstride=0, y2_coff=0)
        ms = time_it(lambda: _capi.check(L.step_conv_forward(ctypes.byref(d), _lib.dptr(x), _lib.dptr(wp), _lib.dptr(sc), _lib.dptr(sh), None, _lib.dptr(y), None, st), name), a.iters)
        gf = 2.0 * B * D * H * W * co * ci * k ** 3 / 1e9
        byts = (x.numel() + y.numel() + wp.numel()) * x.element_size()
        kn = ctypes.create_string_buffer(256)
        L.step_conv_kernel_name(ctypes.byref(d), kn, 256)
        print("%-8s %8.3f ms %8.1f TFLOP/s %8.1f GB/s  %s" % (name, ms, gf / ms, byts / ms / 1e6, kn.value.decode()[11:60]))
        tot_ms += ms
        tot_gf += gf
    print("listed layers: %.3f ms, %.1f GFLOP, %.1f TFLOP/s" % (tot_ms, tot_gf, tot_gf / tot_ms))


if __name__ == "__main__":
    main()
