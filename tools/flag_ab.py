"""tools/flag_ab.py -- Python-level switches of the backbone A/B on the whole C2 step, one / two / three batches in flight, variants interleaved.
   python tools/flag_ab.py "" "ops.POOL_CONV_MAX_NB=3" "opt:conv_tail=0" """
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from step_amd import _capi, _lib, backbone, ops  # noqa: E402


def apply(v):
    saved = []
    for kv in [s for s in v.split(",") if s]:
        k, val = kv.split("=")
        if k.startswith("opt:"):
            _capi.set_option(_lib.lib(), k[4:], int(val))
            saved.append(("opt", None, None))
        else:
            mod, name = k.split(".")
            m = {"ops": ops, "bb": backbone}[mod]
            saved.append((m, name, getattr(m, name)))
            setattr(m, name, type(getattr(m, name))(int(val)))
    return saved


def restore(saved):
    _lib.lib().step_reset_options()
    for m, name, old in saved:
        if m != "opt":
            setattr(m, name, old)


def main():
    variants = sys.argv[1:] or [""]
    NF = 3
    dev = torch.device("cuda:0")
    net = bench.build_net(dev)
    xs = [(torch.rand(8, 32, 3, 224, 224) * 2 - 1).to(dev).to(torch.bfloat16) for _ in range(NF)]
    streams = [torch.cuda.Stream(dev) for _ in range(NF)]
    caps = {}
    with torch.no_grad():
        for v in variants:
            sv = apply(v)
            gs = []
            for b in range(NF):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.stream(streams[b]):
                    for _ in range(2):
                        net(xs[b])
                    torch.cuda.synchronize()
                    with torch.cuda.graph(g, stream=streams[b]):
                        y = net(xs[b])
                gs.append((g, y))
            caps[v] = gs
            restore(sv)
    torch.cuda.synchronize()
    ref = None
    same = {}
    for v in variants:
        caps[v][0][0].replay()
        torch.cuda.synchronize()
        y = caps[v][0][1]
        if ref is None:
            ref = y.clone()
        same[v] = bool(torch.equal(y, ref))

    def run(gs, n, steps=300):
        for _ in range(20):
            gs[0][0].replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            i = k % n
            with torch.cuda.stream(streams[i]):
                gs[i][0].replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3
    res = {v: [[], [], []] for v in variants}
    for _ in range(5):
        for v in variants:
            for n in (1, 2, 3):
                res[v][n - 1].append(run(caps[v], n))
    for v in variants:
        m = [sorted(r)[2] for r in res[v]]
        print("%-34s one %.4f ms = %5.0f | two %.4f ms = %5.0f | three %.4f ms = %5.0f clips/s | same bits: %s" % (
            v or "(default)", m[0], 8 / m[0] * 1e3, m[1], 8 / m[1] * 1e3, m[2], 8 / m[2] * 1e3, same[v]))


if __name__ == "__main__":
    main()
