"""Optimizer pass at the C4 parameter count (44.4 M fp32 parameters in ~250 tensors):
step_amd.optim.FlatAdam (one launch of step_adam_flat) beside torch.optim.Adam over the same single-tensor groups."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import step_amd

dev = torch.device("cuda:0")
torch.manual_seed(0)
# ~50 M parameters in 245 tensors of very different sizes (conv banks down to 256-element biases), like the C4 model
sizes = [64 * 3 * 343, 192 * 64 * 27] + [384 * 192 * 27] * 20 + [256 * 832] * 60 + [1024 * 1024 * 9] * 3 + [256] * 160


def make():
    return [torch.nn.Parameter(torch.randn(n, device=dev) * 0.01) for n in sizes]


def groups(ps):
    return [{"params": [p], "lr": 1e-5 * (1 + i % 3), "weight_decay": 0.0 if i % 2 else 1e-7} for i, p in enumerate(ps)]


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, (time.perf_counter() - t0) * 1e3 / n


a, b = make(), make()
for p in a + b:
    p.grad = torch.randn_like(p)
ot = torch.optim.Adam(groups(a), lr=1e-5)
of = step_amd.FlatAdam(groups(b), lr=1e-5)
n = sum(sizes)
for name, fn, nb in (("torch.optim.Adam (%d groups): zero_grad(set_to_none=False) + step" % len(sizes), lambda: (ot.zero_grad(set_to_none=False), ot.step()), 32),
                     ("FlatAdam: step(zero_grad=True)", lambda: of.step(grad_scale=0.125, zero_grad=True), 32),
                     ("FlatAdam: step()", lambda: of.step(), 28)):
    gpu, wall = timed(fn)
    print("%-75s %8.3f ms device  %8.3f ms wall   %7.1f GB/s (%d B/param)" % (name, gpu, wall, n * nb / gpu / 1e6, nb))
