"""tools/two_in_flight.py -- C2 throughput with ONE captured step replayed back to back against TWO captured steps (two batches, two
streams) replayed alternately, so that the tail of one batch's launches overlaps the other's (GPU only, measurement aid)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    net = bench.build_net(dev)
    NF = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    xs = [(torch.rand(8, 32, 3, 224, 224) * 2 - 1).to(dev).to(torch.bfloat16) for _ in range(NF)]
    streams = [torch.cuda.Stream(dev) for _ in range(NF)]
    graphs = []
    with torch.no_grad():
        net(xs[0]); net(xs[1])
        torch.cuda.synchronize()
        for i in range(NF):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.stream(streams[i]):
                for _ in range(2):
                    net(xs[i])
                torch.cuda.synchronize()
                with torch.cuda.graph(g, stream=streams[i]):
                    y = net(xs[i])
            graphs.append((g, y))
    torch.cuda.synchronize()

    def run(two, steps=200):
        for _ in range(20):
            graphs[0][0].replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            i = (k % NF) if two else 0
            with torch.cuda.stream(streams[i]):
                graphs[i][0].replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3
    for _ in range(3):
        a, b = run(False), run(True)
        print("one batch in flight: %.4f ms/step = %.0f clips/s | %d in flight: %.4f ms/step = %.0f clips/s" % (a, 8 / a * 1e3, NF, b, 8 / b * 1e3))


if __name__ == "__main__":
    main()
