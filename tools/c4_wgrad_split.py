"""tools/c4_wgrad_split.py -- where the weight-gradient time of one C4 training step goes (diagnostic, GPU only):
wraps step_amd.ops.conv_wgrad with HIP events and prints the calls grouped by layer shape."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from step_amd import backbone, ops, workloads  # noqa: E402

dt = {"f32": torch.float32, "bf16": torch.bfloat16}[sys.argv[1] if len(sys.argv) > 1 else "f32"]
w = workloads.C4TrainStep(torch.device("cuda:0"), batch=1, dtype=dt)
for _ in range(2):
    w.step()
backbone.BRANCH_STREAMS = False
REC = []
orig = ops.conv_wgrad


def timed(x, gy, Cout, k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = orig(x, gy, Cout, k)
    e1.record()
    REC.append((tuple(x.shape), Cout, tuple(k), e0, e1))
    return r


ops.conv_wgrad = timed
backbone.ops.conv_wgrad = timed
w.step()
torch.cuda.synchronize()
agg = {}
for shp, co, k, e0, e1 in REC:
    a = agg.setdefault((shp, co, k), [0, 0.0])
    a[0] += 1
    a[1] += e0.elapsed_time(e1)
tot = sum(v[1] for v in agg.values())
print("conv_wgrad: %d calls, %.2f ms total" % (len(REC), tot))
for (shp, co, k), (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    N, D, H, W, ci = shp
    gf = 2.0 * N * D * H * W * ci * co * k[0] * k[1] * k[2] / 1e9 * n
    print("%-28s -> %4d k%s x%d  %7.3f ms  %6.1f TFLOP/s  %4.1f%%" % (shp, co, "%dx%dx%d" % k, n, ms, gf / ms, 100 * ms / tot))
