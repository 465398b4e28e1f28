"""tools/c3_copy_trace.py -- where the C3 step's device-to-device copies come from (torch.profiler with stacks; GPU only, diagnostic)."""
import os
import sys
from collections import Counter

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from step_amd import workloads  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    w = workloads.C3Inference(dev, torch.bfloat16, batch=4, tubes=11, seed=123, graph="--eager" not in sys.argv)
    for _ in range(3):
        w.step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        for _ in range(2):
            w.step()
        torch.cuda.synchronize()
    cnt = Counter()
    for ev in prof.events():
        if ev.name in ("aten::copy_", "aten::_to_copy", "aten::contiguous", "aten::clone", "aten::fill_", "aten::zero_", "aten::index", "aten::gather", "aten::cat"):
            st = [s for s in (ev.stack or []) if "step_amd" in s or "bench" in s or "tools" in s]
            cnt[(ev.name, tuple(st[:2]))] += 1
    for (name, st), n in cnt.most_common(40):
        print(n, name, " <- ".join(s.split("/")[-1] for s in st))
    names = Counter(ev.name for ev in prof.events() if ev.device_type is not None and "DeviceType.CUDA" in str(ev.device_type))
    for n_, c in names.most_common(12):
        print("GPU", c, n_[:100])


if __name__ == "__main__":
    main()
