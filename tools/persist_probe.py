"""tools/persist_probe.py -- timeline of the fused conv3d_2b -> conv3d_2c -> maxPool3d_3a call at C2 size, one workgroup per tile against the
persistent tile loop (option conv_persist), PROBE build of the library (make -C step_amd/csrc PROBE=1; GPU only, tuning aid).

Per workgroup the probe build stores 100 MHz timestamps: entry, first K loop done, exit (all stores acknowledged) and -- persistent form --
the end of each of its tiles.  Reported: per-tile time, the spread of the workgroups' exit times (what a STATIC tile assignment pays when
CUs run at different speeds), idle time of a CU between two consecutive workgroups (one workgroup per tile), span of the launch.

    python tools/persist_probe.py [--batch 8]
"""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from step_amd import _capi  # noqa: E402

PROBE_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libstep_amd_probe.so")
MAXWG = 1 << 15


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--tail", type=int, default=0, help="conv_tail option (0: one launch, no NB = 1 tail launch that would overwrite records)")
    a = ap.parse_args()
    L = _capi.declare(ctypes.CDLL(PROBE_LIB), strict=False)
    L.step_probe_set.argtypes = [ctypes.c_void_p]
    L.step_probe_set.restype = None
    dev = torch.device("cuda:0")
    N, D, H, W, Cout = a.batch, 16, 56, 56, 192
    g = torch.Generator().manual_seed(5)
    x = torch.relu(torch.randn(N, D, H, W, 64, generator=g)).to(dev).to(torch.bfloat16)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def packed(cout, cin, k):
        w = (torch.randn(cout, cin, k, k, k, generator=g) / (cin * k ** 3) ** 0.5).to(dev)
        n = L.step_conv_packed_elems(cout, cin, k, k, k)
        out = torch.empty(n, dtype=torch.bfloat16, device=dev)
        _capi.check(L.step_conv_pack_weight(ctypes.c_void_p(w.data_ptr()), cout, cin, k, k, k, _capi.BF16, None, ctypes.c_void_p(out.data_ptr()), st), "pack")
        return out
    wa, wb = packed(64, 64, 1), packed(Cout, 64, 3)
    sa, ha = torch.ones(64, device=dev), torch.zeros(64, device=dev)
    sb, hb = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)
    d = _capi.ConvDesc(dtype=_capi.BF16, N=N, D=D, H=H, W=W, Cin=64, Cout=Cout, kd=3, kh=3, kw=3, x_cstride=64, x_coff=0, y_cstride=Cout, y_coff=0,
                       res_cstride=0, res_coff=0, relu=1, split=0, y2_cstride=0, y2_coff=0)
    Hp, Wp = L.step_pool_out_size(H, 3, 2), L.step_pool_out_size(W, 3, 2)
    y = torch.empty(N, D, Hp, Wp, Cout, dtype=torch.bfloat16, device=dev)
    nb = L.step_conv_pre_pool_workspace_bytes(ctypes.byref(d))
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    probe = torch.zeros(MAXWG * 16, dtype=torch.int64, device=dev)

    def run():
        _capi.check(L.step_conv_forward_pre_pool_tiles(ctypes.byref(d), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(wb.data_ptr()), ctypes.c_void_p(sb.data_ptr()),
                                                       ctypes.c_void_p(hb.data_ptr()), ctypes.c_void_p(wa.data_ptr()), ctypes.c_void_p(sa.data_ptr()), ctypes.c_void_p(ha.data_ptr()),
                                                       64, ctypes.c_void_p(y.data_ptr()), ctypes.c_void_p(ws.data_ptr()), nb, st), "pre_pool_tiles")
    ref = None
    for persist in (0, 1):
        with _capi.options(L, conv_persist=persist, conv_tail=a.tail):
            L.step_probe_set(None)
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                run()
            e1.record()
            torch.cuda.synchronize()
            kern_us = e0.elapsed_time(e1) * 50.0
            probe.zero_()
            L.step_probe_set(ctypes.c_void_p(probe.data_ptr()))
            run()
            torch.cuda.synchronize()
            L.step_probe_set(None)
        if ref is None:
            ref = y.clone()
        else:
            print("bit-identical to the one-workgroup-per-tile launch:", bool(torch.equal(ref, y)))
        p = probe.cpu().numpy().reshape(MAXWG, 16)
        p = p[p[:, 0] != 0]
        t = p.astype(np.float64) * 0.01                        # us
        t0 = t[:, 0].min()
        cu = (p[:, 15] >> 32) * 256 + ((p[:, 15] >> 8) & 0xff)
        clk = p[:, 14].astype(np.float64) / np.maximum((p[:, 4] - p[:, 0]).astype(np.float64) * 10.0, 1)
        print("conv_persist=%d: %d workgroups on %d CUs, launch %.1f us (probe run: span %.1f us); shader clock over a workgroup's life median %.2f GHz" % (
            persist, len(p), len(set(cu.tolist())), kern_us, t[:, 4].max() - t0, np.median(clk)))
        start, end = t[:, 0] - t0, t[:, 4] - t0
        print("   entry  min/median/max %.1f %.1f %.1f us | exit min/p5/median/p95/max %.1f %.1f %.1f %.1f %.1f us" % (
            start.min(), np.median(start), start.max(), end.min(), np.percentile(end, 5), np.median(end), np.percentile(end, 95), end.max()))
        if persist:
            ends = t[:, 5:14]
            ntile = (p[:, 5:14] != 0).sum(1)
            first = ends[:, 0] - t[:, 0]
            later = []
            for k in range(1, 9):
                ok = p[:, 5 + k] != 0
                later.append(ends[ok, k] - ends[ok, k - 1])
            later = np.concatenate(later)
            print("   tiles per workgroup min/max %d %d | first tile (entry -> end) mean %.2f us | later tiles (end -> end) mean %.2f us, p5 %.2f, p95 %.2f | first K loop done at %.2f us" % (
                ntile.min(), ntile.max(), first.mean(), later.mean(), np.percentile(later, 5), np.percentile(later, 95), (t[:, 2] - t[:, 0]).mean()))
            per_wg = end - start
            print("   workgroup lifetime min/median/max %.1f %.1f %.1f us (static assignment: the launch ends with the slowest)" % (per_wg.min(), np.median(per_wg), per_wg.max()))
        else:
            life = t[:, 4] - t[:, 0]
            gaps = []
            per_cu = {}
            for i in np.argsort(t[:, 0]):
                per_cu.setdefault(int(cu[i]), []).append(i)
            for ids in per_cu.values():
                for a_, b_ in zip(ids[:-1], ids[1:]):
                    gaps.append(t[b_, 0] - t[a_, 4])
            gaps = np.array(gaps) if gaps else np.zeros(1)
            print("   workgroup entry -> exit mean %.2f us (p5 %.2f, p95 %.2f) | K loop done at %.2f | exit -> next entry on the same CU median %.2f us (p95 %.2f) | per tile incl. gap %.2f us" % (
                life.mean(), np.percentile(life, 5), np.percentile(life, 95), (t[:, 2] - t[:, 0]).mean(), np.median(gaps), np.percentile(gaps, 95), life.mean() + np.median(gaps)))
            last = np.array([t[ids[-1], 4] - t0 for ids in per_cu.values()])
            print("   last exit per CU min/median/max %.1f %.1f %.1f us; workgroups per CU min/max %d %d" % (last.min(), np.median(last), last.max(),
                                                                                                      min(len(v) for v in per_cu.values()), max(len(v) for v in per_cu.values())))


if __name__ == "__main__":
    main()
