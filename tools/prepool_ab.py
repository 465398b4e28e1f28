"""tools/prepool_ab.py -- conv3d_2b -> conv3d_2c -> maxPool3d_3a at a given map shape: (a) the planner's own form (general box where it
prefers one: conv_forward_pre + the stand-alone pool) against (b) general boxes off for this call (4 x 8 x 8 tiles: the fused
conv_forward_pre_pool).  GPU only, measurement aid.   python tools/prepool_ab.py N D H W"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from step_amd import _capi, _lib, ops  # noqa: E402


def main():
    N, D, H, W = (int(v) for v in sys.argv[1:5]) if len(sys.argv) >= 5 else (4, 18, 100, 100)
    dev = torch.device("cuda:0")
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(3)
    x = torch.rand(N, D, H, W, 64, generator=g).to(dev).to(dt)
    wa = (torch.randn(64, 64, 1, 1, 1, generator=g) / 8).to(dev)
    wb = (torch.randn(192, 64, 3, 3, 3, generator=g) / (64 * 27) ** 0.5).to(dev)
    sa, ha = torch.ones(64, device=dev), torch.zeros(64, device=dev)
    sb, hb = torch.ones(192, device=dev), torch.zeros(192, device=dev)
    pa, pb = ops.pack_conv_weight(wa, dt), ops.pack_conv_weight(wb, dt)
    pre = (pa, sa, ha, 64)
    L = _lib.lib()

    def separate():
        y = ops.conv_forward_pre(x, pb, 192, (3, 3, 3), sb, hb, True, pre)
        return ops.maxpool_tf(y, (1, 3, 3), (1, 2, 2))

    def fused():
        return ops.conv_forward_pre_pool(x, pb, 192, (3, 3, 3), sb, hb, True, pre)

    outs, fns = {}, {}
    for name, opts, fn in (("planner (separate pool)", {}, separate), ("4x8x8 fused", {"conv_gen": 0}, fused), ("4x8x8 separate", {"conv_gen": 0}, separate)):
        with _capi.options(L, **opts):
            y = fn()
            if y is None:
                print(name, "-> no fused form at this shape"); continue
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                y = fn()
        fns[name] = gr
        outs[name] = y
    torch.cuda.synchronize()
    res = {k: [] for k in fns}
    for _ in range(5):
        for k, gr in fns.items():
            for _ in range(3):
                gr.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                gr.replay()
            torch.cuda.synchronize()
            res[k].append((time.perf_counter() - t0) / 20 * 1e6)
    ref = next(iter(outs.values()))
    for k in fns:
        print("%-26s %8.1f us   same bits as the first: %s" % (k, sorted(res[k])[2], bool(torch.equal(outs[k], ref))))


if __name__ == "__main__":
    main()
