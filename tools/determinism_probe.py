"""tools/determinism_probe.py -- which gradients of the C4 training step are bit-reproducible run to run (GPU only, tuning aid).

    python tools/determinism_probe.py [--dtype fp32|bf16] [--batch 1] [--no-ws]

Runs forward_backward three times on the same weights and clip, compares the gradient arena bit for bit, lists the tensors that
differ (name, relative L2 of the difference) and times the step with the workspace forms of the weight gradients on / off.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="fp32")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--no-ws", action="store_true")
    a = ap.parse_args()
    from step_amd import ops, workloads

    if a.no_ws:
        ops.WGRAD_WS = False
    dev = torch.device("cuda:0")
    dt = torch.float32 if a.dtype == "fp32" else torch.bfloat16
    w = workloads.C4TrainStep(dev, batch=a.batch, seed=123, dtype=dt)
    names = {}
    for mi, m in enumerate(w.mods):
        for k, p in m.named_parameters():
            names[id(p)] = "%d.%s" % (mi, k)
    grads, losses = [], []
    for _ in range(3):
        w.opt.zero_grad()
        losses.append(float(w.forward_backward()))
        torch.cuda.synchronize()
        grads.append(w.opt.flat_grad.clone())
    print("losses", losses)
    for i in (1, 2):
        d = (grads[i] - grads[0])
        print("run %d vs run 0: bit-identical %s, relative L2 %.3e" % (i, bool(torch.equal(grads[i], grads[0])),
                                                                      float(d.double().norm() / grads[0].double().norm())))
    bad = []
    for e in w.opt._entries:
        p, off, n = e[1], int(e[2]), int(e[3])
        a0, a1 = grads[0][off:off + n], grads[1][off:off + n]
        if not torch.equal(a0, a1):
            bad.append((float((a1 - a0).double().norm() / max(float(a0.double().norm()), 1e-30)), names.get(id(p), "?"), n))
    print("%d of %d tensors differ" % (len(bad), len(w.opt._entries)))
    for r, n, k in sorted(bad, reverse=True)[:40]:
        print("  %-60s n=%-9d rel %.3e" % (n, k, r))
    from step_amd import _capi, _lib
    for gather in (1, 0, 1, 0):
        w.nets["roi_net"].pool_layer.deterministic = bool(gather)      # (a per-call argument of the C ABI since ABI 27)
        if True:
            for _ in range(2):
                w._eager_step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                w._eager_step()
            torch.cuda.synchronize()
            print("roi_bwd_gather=%d: %.2f ms / eager step" % (gather, (time.perf_counter() - t0) * 100))
    for ws in ((True, False) if not a.no_ws else (False,)):
        ops.WGRAD_WS = ws
        for _ in range(3):
            w._eager_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            w._eager_step()
        torch.cuda.synchronize()
        print("WGRAD_WS=%s: %.2f ms / eager step" % (ws, (time.perf_counter() - t0) * 100))


if __name__ == "__main__":
    main()
