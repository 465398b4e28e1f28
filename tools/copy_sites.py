"""tools/copy_sites.py -- which lines of step_amd issue the device-to-device copies / casts / clones of one C3 inference step
(TorchDispatchMode over one eager step; GPU only, diagnostic).  python tools/copy_sites.py [tubes | c4]"""
import os
import sys
import traceback
from collections import Counter

import torch
from torch.utils._python_dispatch import TorchDispatchMode

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from step_amd import workloads  # noqa: E402

WATCH = ("aten.copy_", "aten.clone", "aten._to_copy", "aten.contiguous", "aten.cat", "aten.index", "aten.gather", "aten.index_select", "aten.fill_", "aten.zero_",
         "aten.index_add_", "aten.index_add", "aten.index_put_", "aten._index_put_impl_", "aten.scatter_add_", "aten.scatter_add", "aten.zeros", "aten.zeros_like",
         "aten.new_zeros", "aten.add", "aten.add_", "aten.sum", "aten.mul", "aten.embedding_dense_backward")


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.cnt = Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func).rsplit(".", 1)[0]
        if name in WATCH:
            st = [f for f in traceback.extract_stack() if "/step_amd/" in f.filename][-2:]
            nbytes = 0
            for a in args:
                if isinstance(a, torch.Tensor):
                    nbytes = max(nbytes, a.numel() * a.element_size())
            shapes = ",".join(str(tuple(a.shape)) for a in args if isinstance(a, torch.Tensor))[:60]
            self.cnt[(name, " <- ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in reversed(st)) + " " + shapes, nbytes)] += 1
        return func(*args, **(kwargs or {}))


def main():
    dev = torch.device("cuda:0")
    if len(sys.argv) > 1 and sys.argv[1] == "c4":                 # the fixed-tube training step, eager (forward + backward + Adam)
        w = workloads.C4TrainStep(dev, batch=1, dtype=torch.bfloat16)
        step = w._eager_step if hasattr(w, "_eager_step") else w.step
    else:
        tubes = int(sys.argv[1]) if len(sys.argv) > 1 else 34
        w = workloads.C3Inference(dev, torch.bfloat16, batch=4, tubes=tubes, seed=123, graph=False)
        step = w.step
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    with Log() as lg:
        step()
    torch.cuda.synchronize()
    for (name, site, nb), n in sorted(lg.cnt.items(), key=lambda kv: -kv[0][2] * kv[1]):
        print("%3d x %-16s %10d B  %s" % (n, name, nb, site))


if __name__ == "__main__":
    main()
