"""tools/power_probe.py -- is the C2 step limited by the chip's power management rather than by issue slots?  (GPU only, tuning aid)

The same captured step (same kernels, same launch shapes, same instruction streams) is replayed on three data sets, interleaved in one process:
  random   bench.py's weights and U(-1,1) clips
  zeros    all-zero clips and weights (every operand bit constant: the matrix pipe, LDS and the register file toggle almost nothing)
  const    clips and weights filled with one constant (operands constant, products non-zero)
MI355X lowers its clock under matrix load to stay inside its power budget (MI355X_MICROARCH.md, DVFS give-back: the same GEMM binary ran
+15...21 % on zero-filled operands).  If the step is much faster on constant data, its time on real data is set by ENERGY per clip --
wasted matrix work, LDS traffic, data movement -- and hiding latency (prologues, gaps) returns little; if not, by issue slots and latency.
Effective clock beside each loop: step_clock_sample (s_memtime / s_memrealtime).
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    sets = {}
    for name in ("random", "zeros", "const"):
        net = bench.build_net(dev)
        x = [(torch.rand(8, 32, 3, 224, 224) * 2 - 1).to(dev).to(torch.bfloat16) for _ in range(2)]
        if name != "random":
            v = 0.0 if name == "zeros" else 0.5
            with torch.no_grad():
                for m in net.modules():
                    if isinstance(m, torch.nn.Conv3d):
                        m.weight.fill_(v / 16)
                for t in x:
                    t.fill_(v)
        streams = [torch.cuda.current_stream(dev), torch.cuda.Stream(dev)]
        gs = []
        with torch.no_grad():
            for b in range(2):
                with torch.cuda.stream(streams[b]):
                    for _ in range(2):
                        net(x[b])
                    torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                if b == 0:
                    with torch.cuda.graph(g):
                        y = net(x[b])
                else:
                    with torch.cuda.stream(streams[b]), torch.cuda.graph(g, stream=streams[b]):
                        y = net(x[b])
                gs.append((g, y))
        sets[name] = (net, x, streams, gs)
    torch.cuda.synchronize()

    def run(name, two, steps=400):
        net, x, streams, gs = sets[name]
        ec = bench.EffClock(dev)
        for _ in range(20):
            gs[0][0].replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            if k % 8 == 0:
                ec.sample()
            i = (k % 2) if two else 0
            with torch.cuda.stream(streams[i]):
                gs[i][0].replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3, ec.result()
    res = {n: [[], []] for n in sets}
    clk = {n: [None, None] for n in sets}
    for _ in range(3):
        for n in sets:
            for two in (0, 1):
                ms, c = run(n, bool(two))
                res[n][two].append(ms)
                clk[n][two] = c
    for n in sets:
        o, t = sorted(res[n][0])[1], sorted(res[n][1])[1]
        print("%-7s one %.4f ms = %5.0f clips/s (clock %s) | two %.4f ms = %5.0f clips/s (clock %s)" % (
            n, o, 8 / o * 1e3, clk[n][0] and clk[n][0]["ghz_median"], t, 8 / t * 1e3, clk[n][1] and clk[n][1]["ghz_median"]))


if __name__ == "__main__":
    main()
