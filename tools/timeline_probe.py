"""tools/timeline_probe.py -- where a conv_tap launch spends its time, per workgroup and per CU (GPU only, tuning aid).

Uses the PROBE build of the library (make -C step_amd/csrc PROBE=1 -> tools/libstep_amd_probe.so; never the product library):
wave 0 of every workgroup stores the 100 MHz real-time counter at kernel entry, after the prologue (first halo + weights in LDS),
after the K loop, after the last output store was issued and after every store was acknowledged, plus the hardware ids of its CU.
From those: mean phase lengths, the gap a CU idles between two consecutive workgroups, and what fraction of the launch's span the
K loops cover.

    python tools/timeline_probe.py [--batch 8] [--only 2c_3x3,3c_b1b] [--set c2|c3]
"""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from step_amd import _capi  # noqa: E402
from tools.ab_bench import C2, C3  # noqa: E402

PROBE_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libstep_amd_probe.so")
MAXWG = 1 << 17


def report(L, name, run, probe, stem=False):
    L.step_probe_set(None)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    kern_us = e0.elapsed_time(e1) * 100.0
    probe.zero_()
    L.step_probe_set(ctypes.c_void_p(probe.data_ptr()))
    run()                                   # (a layer with a tail launch overwrites the first blocks' records: run with conv_tail=0 for those)
    torch.cuda.synchronize()
    L.step_probe_set(None)
    p = probe.cpu().numpy().reshape(MAXWG, 16)
    p = p[p[:, 0] != 0]
    t = p[:, :5].astype(np.float64) * 0.01                 # us
    cu = (p[:, 15] >> 32) * 256 + ((p[:, 15] >> 8) & 0xff)    # (xcc, se/sh/cu bits of HW_ID)
    t0 = t[:, 0].min()
    span = t[:, 4].max() - t0
    gaps = []
    per_cu = {}
    for i in np.argsort(t[:, 0]):
        per_cu.setdefault(int(cu[i]), []).append(i)
    for ids in per_cu.values():
        for a_, b_ in zip(ids[:-1], ids[1:]):
            gaps.append(t[b_, 0] - t[a_, 3])
    gaps = np.array(gaps) if gaps else np.zeros(1)
    print("%-8s %5d %4d %6.2f | %7.2f %7.2f %7.2f %7.2f %7.2f | %7.2f %6d | %7.1f %7.1f %5.1f%%" % (
        name, len(p), len(per_cu), len(p) / len(per_cu), (t[:, 1] - t[:, 0]).mean(), (t[:, 2] - t[:, 1]).mean(),
        (t[:, 3] - t[:, 2]).mean(), (t[:, 4] - t[:, 3]).mean(), (t[:, 4] - t[:, 0]).mean(), np.median(gaps), len(gaps), span, kern_us,
        100.0 * (t[:, 2] - t[:, 1]).sum() / (span * len(per_cu))))
    f = p[:, :12].astype(np.float64) * 0.01
    if stem:
        print("         K loop: frames 0-1 %.2f, 2-3 %.2f, 4-5 %.2f, 6 %.2f" % (
            (f[:, 5] - f[:, 1]).mean(), (f[:, 6] - f[:, 5]).mean(), (f[:, 7] - f[:, 6]).mean(), (f[:, 2] - f[:, 7]).mean()))
    else:
        f14 = p[:, :15].astype(np.float64) * 0.01
        ok = (f14[:, 7] > 0) & (f14[:, 13] > 0)
        if ok.any():
            g = f14[ok]
            print("         steps 1-3 of slab 0, group 0: MFMA issue %.2f %.2f %.2f us | barrier + load phase + barrier %.2f %.2f %.2f us" % (
                (g[:, 8] - g[:, 7]).mean(), (g[:, 10] - g[:, 9]).mean(), (g[:, 12] - g[:, 11]).mean(),
                (g[:, 9] - g[:, 8]).mean(), (g[:, 11] - g[:, 10]).mean(), (g[:, 13] - g[:, 12]).mean()))
        print("         prologue: weight requests + index tables %.2f, halo load -> LDS %.2f, weights -> LDS + barrier %.2f" % (
            (f[:, 5] - f[:, 0]).mean(), (f[:, 6] - f[:, 5]).mean(), (f[:, 1] - f[:, 6]).mean()))
    cyc = p[:, 14].astype(np.float64)
    okc = (cyc > 0) & (cyc < 1e9) & (t[:, 4] > t[:, 0])
    if okc.any():
        ghz = cyc[okc] / ((t[okc, 4] - t[okc, 0]) * 1e3)
        print("         shader clock over a workgroup's life (s_memtime / s_memrealtime): median %.2f GHz, p5 %.2f, p95 %.2f" % (
            np.median(ghz), np.percentile(ghz, 5), np.percentile(ghz, 95)))
    # first / last start and end spread: how synchronised the CUs are
    firsts = np.array([t[ids[0], 0] for ids in per_cu.values()]) - t0
    ends = np.array([t[ids[-1], 4] for ids in per_cu.values()]) - t0
    print("         first-workgroup entry per CU: %.2f .. %.2f us; last-workgroup end per CU: p5 %.1f median %.1f max %.1f us" % (
        firsts.min(), firsts.max(), np.percentile(ends, 5), np.median(ends), ends.max()))



def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--set", default="c2")
    ap.add_argument("--only", default="2c_3x3,3b_b1b,3c_b1b,4b_b1b,4f_b1b")
    ap.add_argument("--stem", action="store_true", help="the stem (stem_stream_kernel) instead of the conv_tap layers")
    ap.add_argument("--var", default="", help="option=VAL[,option=VAL] planner options")
    a = ap.parse_args()
    L = _capi.declare(ctypes.CDLL(PROBE_LIB))
    L.step_probe_set.argtypes = [ctypes.c_void_p]
    L.step_probe_set.restype = None
    for kv in a.var.split(","):
        if kv:
            k, v = kv.split("=")
            _capi.set_option(L, k, int(v))
    dev = torch.device("cuda:0")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    dt, tdt = _capi.BF16, torch.bfloat16
    B = a.batch
    only = set(x for x in a.only.split(",") if x)
    probe = torch.zeros(MAXWG * 16, dtype=torch.int64, device=dev)
    print("all times in us (100 MHz counter: 0.01 us resolution); gap = CU idle between one workgroup's last store issue and the next one's entry")
    print("%-8s %5s %4s %6s | %7s %7s %7s %7s %7s | %7s %6s | %7s %7s %6s" % (
        "layer", "wgs", "cus", "wg/cu", "prolog", "loop", "epilog", "drain", "total", "gap", "gaps", "span", "kernel", "loop%"))
    if a.stem:
        T_, HW = (32, 224) if a.set == "c2" else (36, 400)
        x = torch.randn(B, T_, 3, HW, HW, device=dev).to(tdt)
        w = torch.randn(64, 3, 7, 7, 7, device=dev) * 0.03
        wp = torch.empty(L.step_stem_packed_elems(64), dtype=tdt, device=dev)
        _capi.check(L.step_stem_pack_weight(ctypes.c_void_p(w.data_ptr()), 64, dt, ctypes.c_void_p(wp.data_ptr()), st), "pack")
        sc, sh = torch.ones(64, device=dev), torch.zeros(64, device=dev)
        y = torch.empty(B, T_ // 2, HW // 2, HW // 2, 64, dtype=tdt, device=dev)

        def run():
            _capi.check(L.step_stem_forward(dt, ctypes.c_void_p(x.data_ptr()), B, T_, HW, HW, ctypes.c_void_p(wp.data_ptr()),
                                            ctypes.c_void_p(sc.data_ptr()), ctypes.c_void_p(sh.data_ptr()), 1, 64,
                                            ctypes.c_void_p(y.data_ptr()), 64, 0, st), "stem")
        report(L, "stem", run, probe, stem=True)
        return
    for name, ci, co, k, D, H, W in (C2 if a.set == "c2" else C3):
        if only and name not in only:
            continue
        x = torch.randn(B, D, H, W, ci, device=dev).to(tdt)
        w = torch.randn(co, ci, k, k, k, device=dev) * (1.0 / (ci * k ** 3) ** 0.5)
        wp = torch.empty(L.step_conv_packed_elems(co, ci, k, k, k), dtype=tdt, device=dev)
        _capi.check(L.step_conv_pack_weight(ctypes.c_void_p(w.data_ptr()), co, ci, k, k, k, dt, None, ctypes.c_void_p(wp.data_ptr()), st), "pack")
        sc, sh = torch.ones(co, device=dev), torch.zeros(co, device=dev)
        y = torch.empty(B, D, H, W, co, dtype=tdt, device=dev)
        d = _capi.ConvDesc(dtype=dt, N=B, D=D, H=H, W=W, Cin=ci, Cout=co, kd=k, kh=k, kw=k, x_cstride=ci, x_coff=0,
                           y_cstride=co, y_coff=0, res_cstride=0, res_coff=0, relu=1, split=0, y2_cstride=0, y2_coff=0)

        def run():
            _capi.check(L.step_conv_forward(ctypes.byref(d), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(wp.data_ptr()),
                                            ctypes.c_void_p(sc.data_ptr()), ctypes.c_void_p(sh.data_ptr()), None,
                                            ctypes.c_void_p(y.data_ptr()), None, st), name)

        report(L, name, run, probe)


if __name__ == "__main__":
    main()
