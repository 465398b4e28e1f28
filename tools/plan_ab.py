"""tools/plan_ab.py -- planner options A/B on the whole C2 step, ONE batch at a time and TWO batches in flight, variants interleaved in one
process (GPU only, measurement aid).   python tools/plan_ab.py "" "conv_nb_rule=1" "conv_nb_rule=1,conv_gen=0" ..."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from step_amd import _capi, _lib  # noqa: E402


def main():
    variants = sys.argv[1:] or ["", "conv_nb_rule=1"]
    dev = torch.device("cuda:0")
    net = bench.build_net(dev)
    L = _lib.lib()
    xs = [(torch.rand(8, 32, 3, 224, 224) * 2 - 1).to(dev).to(torch.bfloat16) for _ in range(2)]
    streams = [torch.cuda.Stream(dev) for _ in range(2)]
    caps, outs = {}, {}
    with torch.no_grad():
        for v in variants:
            L.step_reset_options()
            for kv in [s for s in v.split(",") if s]:
                k, val = kv.split("=")
                _capi.set_option(L, k, int(val))
            gs = []
            for i in range(2):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.stream(streams[i]):
                    for _ in range(2):
                        net(xs[i])
                    torch.cuda.synchronize()
                    with torch.cuda.graph(g, stream=streams[i]):
                        y = net(xs[i])
                gs.append((g, y))
            caps[v] = gs
        L.step_reset_options()
    torch.cuda.synchronize()
    ref = None
    for v in variants:
        caps[v][0][0].replay()
        torch.cuda.synchronize()
        y = caps[v][0][1].float()
        if ref is None:
            ref = y.clone()
        outs[v] = bool(torch.equal(y, ref))

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
    for _ in range(4):
        for v in variants:
            res[v][0].append(run(caps[v], False))
            res[v][1].append(run(caps[v], True))
    for v in variants:
        a, b = sorted(res[v][0])[1], sorted(res[v][1])[1]
        print("%-40s one %.4f ms = %5.0f clips/s | two %.4f ms = %5.0f clips/s | identical to first: %s" % (v or "(default)", a, 8 / a * 1e3, b, 8 / b * 1e3, outs[v]))


if __name__ == "__main__":
    main()
