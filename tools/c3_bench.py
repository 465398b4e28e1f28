"""tools/c3_bench.py -- BASELINE config C3 (full STEP two_branch inference: I3D backbone + ContextNet + 3 refinement
steps with ROIAlign over tubes + batched per-class NMS) and C4 (one training step) on one GPU.  Diagnostic
numbers for DESIGN.md; the headline metric stays bench.py (C2)."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import step_amd  # noqa: E402,F401
from step_amd import workloads  # noqa: E402
from step_amd.driver import GraphedInference, inference, postprocess  # noqa: E402
from step_amd.tube_math import generate_anchors  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--tubes", type=int, default=11)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--train", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    tdt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[a.dtype]
    args, base, ctx, nets = workloads.build_nets(dev)
    x = (torch.rand(a.batch, 36, 3, 400, 400, device=dev) * 2 - 1).to(tdt)
    anchors = generate_anchors()[:a.tubes] * 400.0
    tubes = [np.tile(anchors[:, None, :], (1, 3, 1)).astype(np.float32) for _ in range(a.batch)]

    def run():
        with torch.no_grad():
            cf = base(x)
            cx = ctx(cf)
            hist, _ = inference(args, cf, cx, nets, 3, tubes)
            return postprocess(args, hist)

    out = run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        out = run()
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / a.iters
    kept = sum(int(d_["scores"].shape[0]) for it_ in out for d_ in it_)       # postprocess: iterations x clips x {boxes, scores, labels, tubes}
    print("C3 inference: batch %d x [36,3,400,400] %s, %d tubes/clip: %.2f ms/batch = %.1f clips/s  (detections kept: %d)"
          % (a.batch, a.dtype, a.tubes, el * 1e3, a.batch / el, kept))
    # the same pipeline captured in a hipGraph
    gi = GraphedInference(args, base, ctx, nets, x, tubes)
    with torch.no_grad():
        h2, _, _ = gi(x)
        ref_hist = inference(args, base(x), ctx(base(x)), nets, 3, tubes)[0]
        err = max(float((a["pred_loc"].float() - b["pred_loc"].float()).abs().max()) for a, b in zip(h2, ref_hist))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            h2, _, _ = gi(x)
            out = postprocess(args, h2)
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / a.iters
    print("C3 inference, hipGraph replay + postprocess: %.2f ms/batch = %.1f clips/s  (max |pred_loc diff| vs eager %.3g)"
          % (el * 1e3, a.batch / el, err))
    # stage split
    with torch.no_grad():
        cf0 = base(x)
        cx0 = ctx(cf0)
        from step_amd.driver import _flat_tubes
        flat0, nums0 = _flat_tubes(tubes, dev)
        pooled3 = nets["roi_net"](cf0[:, 3:6], flat0).reshape(-1, 3, 832, 7, 7)
        flat9 = torch.cat([flat0, flat0, flat0], 1).clone()
        flat9[:, :, 0] = (flat0[:, :1, 0] // 3 * 9) + torch.arange(9, device=dev).view(1, 9)
        pooled9 = nets["roi_net"](cf0, flat9).reshape(-1, 9, 832, 7, 7)
        clip_of = torch.as_tensor(np.repeat(np.arange(len(nums0)), nums0), device=dev)
        c3_ = cx0[clip_of][:, :, 3:6]
        c9_ = cx0[clip_of]
        for name, fn in (("backbone", lambda: base(x)), ("context", lambda: ctx(cf0)),
                         ("roialign T3", lambda: nets["roi_net"](cf0[:, 3:6], flat0)),
                         ("roialign T9", lambda: nets["roi_net"](cf0, flat9)),
                         ("head T=3", lambda: nets["det_net0"](pooled3, context_feat=c3_)),
                         ("head T=9", lambda: nets["det_net2"](pooled9, context_feat=c9_))):
            fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(a.iters):
                fn()
            torch.cuda.synchronize()
            print("   %-10s %.2f ms/batch" % (name, (time.perf_counter() - t0) / a.iters * 1e3))
    if a.train:
        # C4: one training step (fp32) of backbone + context + 3 heads on this GPU -- step_amd.workloads.C4TrainStep
        del base, ctx, nets
        w = workloads.C4TrainStep(dev, batch=1)
        t0 = time.perf_counter()
        for it in range(4):
            if it == 2:                                   # two warm-up steps (weight packs, allocator)
                torch.cuda.synchronize()
                print("   (2 warm-up training steps: %.1f s)" % (time.perf_counter() - t0))
                t0 = time.perf_counter()
            loss = w.step()
        torch.cuda.synchronize()
        print("C4 training step (1 clip, fp32, 3 heads): %.1f ms, loss %.4f, grads finite: %s"
              % ((time.perf_counter() - t0) / 2 * 1e3, float(loss), all(bool(torch.isfinite(p.grad).all()) for p in w.params if p.grad is not None)))


if __name__ == "__main__":
    main()
