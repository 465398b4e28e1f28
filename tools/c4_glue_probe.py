"""tools/c4_glue_probe.py -- which Python lines launch the torch glue kernels (add / copy / fill / index_add / sum) that remain in the C4 training step
(GPU only, tuning aid).  One eager bf16 step under torch.profiler with Python stacks; aten ops that launch a device kernel are grouped by the innermost
step_amd / workloads frame of their stack."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from step_amd import workloads  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    w = workloads.C4TrainStep(dev, batch=1, tubes_per_clip=5, seed=123, dtype=torch.bfloat16)
    for _ in range(3):
        w._eager_step()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        w._eager_step()
        torch.cuda.synchronize()
    want = ("aten::add_", "aten::add", "aten::copy_", "aten::fill_", "aten::zero_", "aten::index_add_", "aten::sum", "aten::index_select", "aten::mul", "aten::cat",
            "aten::_to_copy", "aten::clone", "aten::contiguous", "aten::zeros", "aten::zeros_like", "aten::index", "aten::mean", "aten::div", "aten::sub", "aten::neg")
    agg = collections.Counter()
    tim = collections.Counter()
    for e in prof.events():
        if e.name not in want or e.device_time_total <= 0:
            continue
        frame = "?"
        for fr in (e.stack or []):
            if "step_amd/" in fr or "workloads" in fr:
                frame = fr.split("/")[-1]
                break
        shapes = str(e.input_shapes)[:60]
        agg[(e.name, frame, shapes)] += 1
        tim[(e.name, frame, shapes)] += e.device_time_total
    tot = sum(tim.values())
    print("device time of the listed aten ops: %.1f us over %d calls" % (tot, sum(agg.values())))
    for k, t in tim.most_common(40):
        print("%8.1f us %4d x  %-18s %-42s %s" % (t, agg[k], k[0], k[1], k[2]))


if __name__ == "__main__":
    main()
