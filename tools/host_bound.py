"""tools/host_bound.py -- is the C4 training step bound by the host (Python + launch calls) or by the GPU?  Prints, per step, the
time until step() returns (every launch issued) and the time until the GPU has finished (GPU only, tuning aid)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from step_amd import workloads  # noqa: E402

dev = torch.device("cuda:0")
dt = {"bf16": torch.bfloat16, "f32": torch.float32}[sys.argv[1] if len(sys.argv) > 1 else "bf16"]
w = workloads.C4TrainStep(dev, batch=1, seed=123, dtype=dt)
for _ in range(4):
    w.step()
torch.cuda.synchronize()
hs, ts = [], []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    w.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    hs.append((t1 - t0) * 1e3)
    ts.append((t2 - t0) * 1e3)
hs.sort(); ts.sort()
print("host issue time %.2f ms (median), step until GPU idle %.2f ms (median)" % (hs[len(hs) // 2], ts[len(ts) // 2]))
