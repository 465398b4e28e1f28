"""tools/cpu_baselines.py -- the CPU figures SURVEY.md sec. 8(d) / BASELINE.md sec. 3 name, measured on the host cores of the box this
runs on (the GPU box): the torch-CPU fp32 restatement (oracle/i3d_ref.py, test infrastructure) of
  C1  BaseNet on [1,8,3,112,112]
  C2  BaseNet on the C2 clip shape at batch 1 and batch 8
  C3  BaseNet + ContextNet + 3-step inference() + post-processing at B = 1 with 11 and with 34 tubes
1 warm-up + up to 5 timed iterations (bounded by --budget seconds each), median; thread count = the fastest of a probed set
(torch's CPU conv3d stops scaling long before 256 threads: the probe table is printed).  Output: one JSON object per line."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import i3d_ref as R  # noqa: E402
from oracle import postprocess_ref as PR  # noqa: E402


def timed(fn, budget, iters=5):
    fn()
    ts = []
    t_all = time.perf_counter()
    for _ in range(iters):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
        if time.perf_counter() - t_all > budget:
            break
    ts.sort()
    return ts[len(ts) // 2], len(ts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--budget", type=float, default=20.0)
    a = ap.parse_args()
    from step_amd import workloads
    args, base, ctx, nets = workloads.build_nets(torch.device("cpu"))
    f = lambda m: {k: v.detach().float().cpu() for k, v in m.state_dict().items()}
    sd_b, sd_c = f(base), f(ctx)
    nets_sd = {k: f(v) for k, v in nets.items() if k.startswith("det_net")}
    ncpu = os.cpu_count() or 1
    g = torch.Generator().manual_seed(123)
    rnd = lambda *s: torch.rand(*s, generator=g) * 2 - 1
    x1, x2, x3 = rnd(1, 8, 3, 112, 112), rnd(8, 32, 3, 224, 224), rnd(1, 36, 3, 400, 400)
    with torch.no_grad():
        best, table = bench._cpu_threads(lambda: R.basenet_forward(x2[:1, :8], sd_b), ncpu)
        head = {"cpu_model": bench._cpu_model(), "host_cores": ncpu, "threads_used": best, "threads_probe_s (quarter-length C2 clip)": table}
        print(json.dumps(head), flush=True)
        rows = [("C1 BaseNet [1,8,3,112,112]", 1, lambda: R.basenet_forward(x1, sd_b)),
                ("C2 BaseNet [1,32,3,224,224]", 1, lambda: R.basenet_forward(x2[:1], sd_b)),
                ("C2 BaseNet [8,32,3,224,224]", 8, lambda: R.basenet_forward(x2, sd_b))]
        from step_amd.tube_math import generate_anchors
        for tubes in (11, 34):
            anchors = generate_anchors()[:tubes] * 400.0
            tl = [np.tile(anchors[:, None, :], (1, 3, 1)).astype(np.float32)]

            def c3(tl=tl):
                cf = R.basenet_forward(x3, sd_b)
                cx = R.contextnet_forward(cf, sd_c)
                return PR.postprocess(R.inference(cf, cx, nets_sd, tl))
            rows.append(("C3 full inference [1,36,3,400,400], %d tubes" % tubes, 1, c3))
        for name, clips, fn in rows:
            med, n = timed(fn, a.budget)
            print(json.dumps({"config": name, "clips_per_s": round(clips / med, 4), "s_per_iteration_median": round(med, 4), "iterations": n,
                              "threads": torch.get_num_threads()}), flush=True)


if __name__ == "__main__":
    main()
