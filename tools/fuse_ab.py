"""tools/fuse_ab.py -- backbone fusion flags A/B on the whole C2 step (one batch at a time, two in flight), variants interleaved in one process."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from step_amd import backbone as bb  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    net = bench.build_net(dev)
    xs = [(torch.rand(8, 32, 3, 224, 224) * 2 - 1).to(dev).to(torch.bfloat16) for _ in range(2)]
    streams = [torch.cuda.Stream(dev) for _ in range(2)]
    variants = {"conv_pool_fused": True, "separate_pool": False}
    caps, outs = {}, {}
    with torch.no_grad():
        for name, flag in variants.items():
            bb.FUSE_CONV_POOL = flag
            gs = []
            for b in range(2):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.stream(streams[b]):
                    for _ in range(2):
                        net(xs[b])
                    torch.cuda.synchronize()
                    with torch.cuda.graph(g, stream=streams[b]):
                        y = net(xs[b])
                gs.append((g, y))
            caps[name] = gs
        bb.FUSE_CONV_POOL = True
    torch.cuda.synchronize()
    for name in variants:
        caps[name][0][0].replay()
        torch.cuda.synchronize()
        outs[name] = caps[name][0][1].clone()
    print("bit-identical:", bool(torch.equal(outs["conv_pool_fused"], outs["separate_pool"])))

    def run(gs, two, steps=300):
        for _ in range(20):
            gs[0][0].replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            i = (k % 2) if two else 0
            with torch.cuda.stream(streams[i]):
                gs[i][0].replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3
    res = {v: [[], []] for v in variants}
    for _ in range(5):
        for v in variants:
            res[v][0].append(run(caps[v], False))
            res[v][1].append(run(caps[v], True))
    for v in variants:
        a, b = sorted(res[v][0])[2], sorted(res[v][1])[2]
        print("%-18s one %.4f ms = %5.0f clips/s | two %.4f ms = %5.0f clips/s" % (v, a, 8 / a * 1e3, b, 8 / b * 1e3))


if __name__ == "__main__":
    main()
