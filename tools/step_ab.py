"""tools/step_ab.py -- whole-step A/B of library builds and / or planner options on the C2 (or C5) backbone step, variants interleaved in
ONE process (boxes of the pool differ by 5-10 %: only same-process A/Bs are trusted).

    python tools/step_ab.py --var default --var lib=prev [--var conv_persist=0,lib=prev] [--config c2|c5] [--rounds 5] [--steps 300]

A variant is a comma list of `option=value` planner options (step_set_option) and at most one `lib=NAME` (tools/libstep_amd_NAME.so: `prev`
from tools/build_prev.sh, experiment builds from `make -C step_amd/csrc EXP=NAME EXPFLAGS=-D...`); `default` = the working tree's library,
default options.  Each variant's step is captured (HIP graph) for two batches under its own library / options, then the variants' graphs are
replayed round-robin: one batch at a time and two in flight.  Outputs are compared with the first variant's (bit-identity is reported, not
required)."""
import argparse
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from step_amd import _capi, _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--var", action="append", default=[])
    ap.add_argument("--config", default="c2", choices=["c2", "c5"])
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--steps", type=int, default=300)
    a = ap.parse_args()
    variants = a.var or ["default", "lib=prev"]
    c = bench.CONFIGS[a.config]
    dev = torch.device("cuda:0")
    tdt = {"bf16": torch.bfloat16, "f16": torch.float16}[c["dtype"]]
    net = bench.build_net(dev)
    xs = [(torch.rand(c["clips"], c["T"], 3, c["HW"], c["HW"]) * 2 - 1).to(dev).to(tdt) for _ in range(2)]
    streams = [torch.cuda.current_stream(dev), torch.cuda.Stream(dev)]     # (batch 0 on the default stream: bench.py's arrangement)
    main_lib = _lib.lib()
    libs = {}

    def parse(v):
        L, opts = main_lib, {}
        for kv in v.split(","):
            if kv == "default" or not kv:
                continue
            k_, v_ = kv.split("=")
            if k_ == "lib":
                if v_ not in libs:
                    libs[v_] = _capi.declare(ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libstep_amd_%s.so" % v_)), strict=False)
                L = libs[v_]
            else:
                opts[k_] = int(v_)
        return L, opts
    caps, outs = {}, {}
    with torch.no_grad():
        net(xs[0])                                               # weight packs (the packed format is the same for every build compared)
        torch.cuda.synchronize()
        for v in variants:
            L, opts = parse(v)
            _lib._LIB = L
            try:
                with _capi.options(L, **opts):
                    gs = []
                    for b in range(2):
                        g = torch.cuda.CUDAGraph()
                        with torch.cuda.stream(streams[b]):
                            for _ in range(2):
                                net(xs[b])
                            torch.cuda.synchronize()
                        if b == 0:                               # (captured on the capture's own side stream, replayed on the default stream)
                            with torch.cuda.graph(g):
                                y = net(xs[b])
                        else:
                            with torch.cuda.stream(streams[b]), torch.cuda.graph(g, stream=streams[b]):
                                y = net(xs[b])
                        gs.append((g, y))
                    caps[v] = gs
            finally:
                _lib._LIB = main_lib
    torch.cuda.synchronize()
    for v in variants:
        caps[v][0][0].replay()
        torch.cuda.synchronize()
        outs[v] = caps[v][0][1].clone()
    for v in variants[1:]:
        same = bool(torch.equal(outs[v], outs[variants[0]]))
        d = float((outs[v].float() - outs[variants[0]].float()).abs().max())
        print("%-40s vs %-20s bit-identical: %s (max abs diff %.3e)" % (v, variants[0], same, d))

    def run(gs, two, steps):
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
    for _ in range(a.rounds):
        for v in variants:
            res[v][0].append(run(caps[v], False, a.steps))
            res[v][1].append(run(caps[v], True, a.steps))
    n = c["clips"]
    for v in variants:
        o, t = sorted(res[v][0]), sorted(res[v][1])
        print("%-40s one %.4f ms (min %.4f) = %5.0f clips/s | two %.4f ms (min %.4f) = %5.0f clips/s" % (
            v, o[len(o) // 2], o[0], n / o[len(o) // 2] * 1e3, t[len(t) // 2], t[0], n / t[len(t) // 2] * 1e3))


if __name__ == "__main__":
    main()
