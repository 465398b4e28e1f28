"""tools/skip_probe.py -- UPPER BOUND of what fusing a stand-alone pool into its producer could buy: the C2 step with maxPool3d_3a and / or
maxPool3d_4a simply NOT LAUNCHED (their consumers read a stale buffer of the right shape: wrong results, right timing of everything
else), one batch at a time and two batches in flight, variants interleaved.  Measurement aid, GPU only."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    net = bench.build_net(dev)
    xs = [(torch.rand(8, 32, 3, 224, 224) * 2 - 1).to(dev).to(torch.bfloat16) for _ in range(2)]
    streams = [torch.cuda.Stream(dev) for _ in range(2)]
    variants = {"baseline": (), "no_pool3a": (4,), "no_pool4a": (7,), "no_pool3a_4a": (4, 7)}
    caps = {}
    with torch.no_grad():
        # the pooled shapes, once
        keep = {}
        hooks = [net.base_model[i].register_forward_hook(lambda m, a, o, i=i: keep.__setitem__(i, o.detach().clone())) for i in (4, 7)]
        net(xs[0])
        for h in hooks:
            h.remove()
        torch.cuda.synchronize()
        orig = {i: net.base_model[i].forward for i in (4, 7)}
        for name, skip in variants.items():
            for i in (4, 7):
                net.base_model[i].forward = (lambda x, i=i: keep[i]) if i in skip else orig[i]
            gs = []
            for b in range(2):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.stream(streams[b]):
                    for _ in range(2):
                        net(xs[b])
                    torch.cuda.synchronize()
                    with torch.cuda.graph(g, stream=streams[b]):
                        net(xs[b])
                gs.append(g)
            caps[name] = gs
        for i in (4, 7):
            net.base_model[i].forward = orig[i]
    torch.cuda.synchronize()

    def run(gs, two, steps=300):
        for _ in range(20):
            gs[0].replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            i = (k % 2) if two else 0
            with torch.cuda.stream(streams[i]):
                gs[i].replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3
    res = {v: [[], []] for v in variants}
    for _ in range(4):
        for v in variants:
            res[v][0].append(run(caps[v], False))
            res[v][1].append(run(caps[v], True))
    for v in variants:
        a, b = sorted(res[v][0])[1], sorted(res[v][1])[1]
        print("%-16s one %.4f ms = %5.0f clips/s | two %.4f ms = %5.0f clips/s" % (v, a, 8 / a * 1e3, b, 8 / b * 1e3))


if __name__ == "__main__":
    main()
