"""tools/ab_bench.py -- interleaved A/B timing of planner variants per layer (GPU only, tuning aid).

Every layer of the C2 backbone (the fused 1x1x1 triples as the net really launches them) is timed under several
planner options (step_set_option: conv_waves=8|4, ...), interleaved in ONE process (round-robin over the
variants, median over rounds) as cdna_hip_programming.md 5.4 rule 24 asks.

    python tools/ab_bench.py [--batch 8] [--rounds 7] [--iters 10] [--set c2|c3] [--var "conv_waves=8" --var "conv_waves=4"]
"""
import argparse
import ctypes
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from step_amd import _capi, _lib  # noqa: E402

# (name, Cin, Cout, k, D, H, W) per clip
C2 = [
    ("2b_1x1", 64, 64, 1, 16, 56, 56), ("2c_3x3", 64, 192, 3, 16, 56, 56),
    ("3b_f1x1", 192, 176, 1, 16, 28, 28), ("3b_b1b", 96, 128, 3, 16, 28, 28), ("3b_b2b", 16, 32, 3, 16, 28, 28), ("3b_b3", 192, 32, 1, 16, 28, 28),
    ("3c_f1x1", 256, 288, 1, 16, 28, 28), ("3c_b1b", 128, 192, 3, 16, 28, 28), ("3c_b2b", 32, 96, 3, 16, 28, 28), ("3c_b3", 256, 64, 1, 16, 28, 28),
    ("4b_f1x1", 480, 304, 1, 8, 14, 14), ("4b_b1b", 96, 208, 3, 8, 14, 14), ("4b_b2b", 16, 48, 3, 8, 14, 14), ("4b_b3", 480, 64, 1, 8, 14, 14),
    ("4c_f1x1", 512, 296, 1, 8, 14, 14), ("4c_b1b", 112, 224, 3, 8, 14, 14), ("4c_b2b", 24, 64, 3, 8, 14, 14),
    ("4d_f1x1", 512, 280, 1, 8, 14, 14), ("4d_b1b", 128, 256, 3, 8, 14, 14),
    ("4e_f1x1", 512, 288, 1, 8, 14, 14), ("4e_b1b", 144, 288, 3, 8, 14, 14), ("4e_b2b", 32, 64, 3, 8, 14, 14),
    ("4f_f1x1", 528, 448, 1, 8, 14, 14), ("4f_b1b", 160, 320, 3, 8, 14, 14), ("4f_b2b", 32, 128, 3, 8, 14, 14), ("4f_b3", 528, 128, 1, 8, 14, 14),
]
# AVA-shaped clip [36,3,400,400]: maps 100x100 (18 planes), 50x50 (18), 25x25 (9)
C3 = [
    ("2c_3x3", 64, 192, 3, 18, 100, 100), ("3b_f1x1", 192, 176, 1, 18, 50, 50), ("3b_b1b", 96, 128, 3, 18, 50, 50),
    ("3c_f1x1", 256, 288, 1, 18, 50, 50), ("3c_b1b", 128, 192, 3, 18, 50, 50), ("3c_b2b", 32, 96, 3, 18, 50, 50),
    ("4b_f1x1", 480, 304, 1, 9, 25, 25), ("4b_b1b", 96, 208, 3, 9, 25, 25), ("4e_b1b", 144, 288, 3, 9, 25, 25), ("4f_b1b", 160, 320, 3, 9, 25, 25),
    ("5b_b1b", 160, 320, 3, 9, 13, 13), ("5c_b1b", 192, 384, 3, 9, 13, 13),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--set", default="c2")
    ap.add_argument("--only", default="")
    ap.add_argument("--custom", action="append", default=[], help="name,Cin,Cout,k,D,H,W (repeatable; replaces the layer set)")
    ap.add_argument("--nocheck", action="store_true", help="variants may compute different things (timing experiments)")
    ap.add_argument("--var", action="append", default=[], help="option=VAL[,option=VAL] planner options of one variant, or `default` (repeatable)")
    a = ap.parse_args()
    variants = a.var or ["conv_waves=8", "conv_waves=4"]
    L = _lib.lib()
    # `lib=NAME` in a variant: tools/libstep_amd_NAME.so -- `prev` from tools/build_prev.sh, experiment builds from
    # `make -C step_amd/csrc EXP=NAME EXPFLAGS=-D...`
    LIBS = {}

    def lib_of(v):
        for kv in v.split(","):
            if kv.startswith("lib="):
                n = kv[4:]
                if n not in LIBS:
                    LIBS[n] = _capi.declare(ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libstep_amd_%s.so" % n)), strict=False)
                return LIBS[n]
        return L
    dt, tdt = _capi.BF16, torch.bfloat16
    dev = torch.device("cuda:0")
    st = _lib.stream_ptr()
    B = a.batch
    layers = C2 if a.set == "c2" else C3
    if a.custom:
        layers = []
        for c in a.custom:
            f = c.split(",")
            layers.append((f[0],) + tuple(int(v) for v in f[1:]))
    only = set(x for x in a.only.split(",") if x)
    tot = [0.0] * len(variants)
    print("%-8s %s" % ("layer", "  ".join("%28s" % v for v in variants)))
    if "stem" in only or a.set == "stem":
        # conv3d_1a_7x7 (its own entry point): B clips [32,3,224,224] (c2) or [36,3,400,400] (c3)
        T_, HW_ = (36, 400) if a.set == "c3" else (32, 224)
        x = (torch.rand(B, T_, 3, HW_, HW_, device=dev) * 2 - 1).to(tdt)
        w = torch.randn(64, 3, 7, 7, 7, device=dev) * (1.0 / 1029 ** 0.5)
        sc, sh = torch.ones(64, device=dev), torch.zeros(64, device=dev)
        To, Ho = (T_ - 2) // 2 + 1, (HW_ - 2) // 2 + 1
        y = torch.empty(B, To, Ho, Ho, 64, dtype=tdt, device=dev)
        times, ref = [[] for _ in variants], None
        libs = [lib_of(v) for v in variants]
        wps = []
        for lib in libs:
            wp = torch.empty(lib.step_stem_packed_elems(64), dtype=tdt, device=dev)
            _capi.check(lib.step_stem_pack_weight(_lib.dptr(w), 64, dt, _lib.dptr(wp), st), "stem pack")
            wps.append(wp)

        def run_stem(vi):
            _capi.check(libs[vi].step_stem_forward(dt, _lib.dptr(x), B, T_, HW_, HW_, _lib.dptr(wps[vi]), _lib.dptr(sc), _lib.dptr(sh), 1, 64,
                                                   _lib.dptr(y), 64, 0, st), "stem")
        for vi in range(len(variants)):
            run_stem(vi)
            torch.cuda.synchronize()
            if ref is None:
                ref = y.float().clone()
            else:
                assert a.nocheck or float((y.float() - ref).abs().max() / ref.abs().max()) < 2e-2
        for _ in range(a.rounds):
            for vi in range(len(variants)):
                run_stem(vi)
                s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s_.record()
                for _ in range(a.iters):
                    run_stem(vi)
                e_.record()
                torch.cuda.synchronize()
                times[vi].append(s_.elapsed_time(e_) / a.iters)
        gf = 2.0 * B * To * Ho * Ho * 64 * 1029 / 1e9
        med = [statistics.median(t) for t in times]
        print("%-8s %s" % ("stem", "  ".join("%7.1f us %6.0f TF %-12s" % (m * 1e3, gf / m, "stem_stream") for m in med)))
        if a.set == "stem":
            return
    for lay in layers:
        name, ci, co, k, D, H, W = lay[:7]
        with_res = len(lay) > 7 and lay[7]                 # --custom name,Cin,Cout,k,D,H,W,1: a residual tensor of the output's shape
        if only and name not in only:
            continue
        x = torch.randn(B, D, H, W, ci, device=dev).to(tdt)
        w = torch.randn(co, ci, k, k, k, device=dev) * (1.0 / (ci * k ** 3) ** 0.5)
        wp = torch.empty(L.step_conv_packed_elems(co, ci, k, k, k), dtype=tdt, device=dev)
        _capi.check(L.step_conv_pack_weight(_lib.dptr(w), co, ci, k, k, k, dt, None, _lib.dptr(wp), st), "pack")
        sc, sh = torch.ones(co, device=dev), torch.zeros(co, device=dev)
        y = torch.empty(B, D, H, W, co, dtype=tdt, device=dev)
        res = torch.randn(B, D, H, W, co, device=dev).to(tdt) if with_res else None
        d = _capi.ConvDesc(dtype=dt, N=B, D=D, H=H, W=W, Cin=ci, Cout=co, kd=k, kh=k, kw=k, x_cstride=ci, x_coff=0,
                           y_cstride=co, y_coff=0, res_cstride=co if with_res else 0, res_coff=0, relu=1, split=0, y2_cstride=0, y2_coff=0)

        cur = [L]

        def run():
            _capi.check(cur[0].step_conv_forward(ctypes.byref(d), _lib.dptr(x), _lib.dptr(wp), _lib.dptr(sc), _lib.dptr(sh), _lib.dptr(res) if with_res else None, _lib.dptr(y), None, st), name)

        def setenv(v):                       # a variant = planner options (include/step_amd.h), everything else at its default
            cur[0] = lib_of(v)
            cur[0].step_reset_options()
            for kv in v.split(","):
                if kv and kv != "default" and not kv.startswith("lib="):
                    kk, vv = kv.split("=")
                    _capi.set_option(cur[0], kk, int(vv))

        times = [[] for _ in variants]
        names = []
        ref = None
        for vi, v in enumerate(variants):
            setenv(v)
            kn = ctypes.create_string_buffer(256)
            cur[0].step_conv_kernel_name(ctypes.byref(d), kn, 256)
            names.append(kn.value.decode()[11:].split("(")[0].replace("step::", "").replace("_kernel", ""))
            run()
            torch.cuda.synchronize()
            if ref is None:
                ref = y.float().clone()
            else:
                err = float((y.float() - ref).abs().max() / ref.abs().max().clamp_min(1e-20))
                assert a.nocheck or err < 2e-2, (name, v, err)            # every variant computes the same layer
        for _ in range(a.rounds):
            for vi, v in enumerate(variants):
                setenv(v)
                run()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(a.iters):
                    run()
                e.record()
                torch.cuda.synchronize()
                times[vi].append(s.elapsed_time(e) / a.iters)
        gf = 2.0 * B * D * H * W * co * ci * k ** 3 / 1e9
        med = [statistics.median(t) for t in times]
        for vi in range(len(variants)):
            tot[vi] += med[vi]
        print("%-8s %s" % (name, "  ".join("%7.1f us %6.0f TF %-12s" % (m * 1e3, gf / m, n[-14:]) for m, n in zip(med, names))))
    print("%-8s %s" % ("total", "  ".join("%7.1f us %21s" % (t * 1e3, "") for t in tot)))
    L.step_reset_options()


if __name__ == "__main__":
    main()
