"""tools/skip_b2b.py -- UPPER BOUND of what a dedicated small-Cin 3x3x3 kernel could buy: the C2 step with the branch_2 3x3x3 members
(Cin 16-32) simply dropped from their grouped launches (wrong results, right timing of everything else); one / two batches in flight."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from step_amd import ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    net = bench.build_net(dev)
    xs = [(torch.rand(8, 32, 3, 224, 224) * 2 - 1).to(dev).to(torch.bfloat16) for _ in range(2)]
    streams = [torch.cuda.current_stream(), torch.cuda.Stream(dev)]
    orig = ops.conv_forward_group
    mode = {"v": "all"}

    def patched(members):
        if mode["v"] == "all":
            return orig(members)
        small = [m for m in members if tuple(m[3]) == (3, 3, 3) and m[0].shape[-1] <= 32 and (mode["v"] != "no_b2b_28" and mode["v"] != "sep_b2b_28" or m[0].shape[2] >= 28)]
        keep = [m for m in members if not any(m is s_ for s_ in small)]
        if mode["v"].startswith("sep"):                        # the small-Cin members as launches of their own (their own plan: the two-workgroups-per-CU NB = 1 form)
            for (x_, w_, co_, k_, sc_, sh_, relu_, out_) in small:
                ops.conv_forward(x_, w_, co_, k_, sc_, sh_, relu_, None, out_)
        if len(keep) == 1:
            x_, w_, co_, k_, sc_, sh_, relu_, out_ = keep[0]
            return ops.conv_forward(x_, w_, co_, k_, sc_, sh_, relu_, None, out_)
        return orig(keep) if keep else None
    ops.conv_forward_group = patched
    caps = {}
    with torch.no_grad():
        for v in ("all", "no_b2b_28", "sep_b2b_28", "sep_b2b"):
            mode["v"] = v
            gs = []
            for b in range(2):
                g = torch.cuda.CUDAGraph()
                for _ in range(2):
                    net(xs[b])
                torch.cuda.synchronize()
                with torch.cuda.graph(g):                          # (captured on torch's capture stream; replayed on the default / the pool stream)
                    net(xs[b])
                gs.append(g)
            caps[v] = gs
    ops.conv_forward_group = orig
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
    res = {v: [[], []] for v in caps}
    for _ in range(4):
        for v in caps:
            res[v][0].append(run(caps[v], False))
            res[v][1].append(run(caps[v], True))
    for v in caps:
        a, b = sorted(res[v][0])[1], sorted(res[v][1])[1]
        print("%-12s one %.4f ms = %5.0f clips/s | two %.4f ms = %5.0f clips/s" % (v, a, 8 / a * 1e3, b, 8 / b * 1e3))


if __name__ == "__main__":
    main()
