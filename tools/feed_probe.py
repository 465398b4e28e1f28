"""tools/feed_probe.py -- variants of the fed C2 loop (bench.py --feed u8), same captured steps, one process:
   resident | staged copy on a normal / high-priority copy stream | 2-4 batches in flight | zero-copy conversion from pinned memory."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from step_amd import ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    net = bench.build_net(dev)
    N, T, HW = 8, 32, 224
    nfl_max = int(os.environ.get("NFL", "4"))
    g = torch.Generator().manual_seed(123)
    flights = []
    with torch.no_grad():
        for i in range(nfl_max):
            xi = (torch.rand(N, T, 3, HW, HW, generator=g) * 2 - 1).to(dev).to(torch.bfloat16)
            si = torch.cuda.Stream()
            with torch.cuda.stream(si):
                for _ in range(2):
                    net(xi)
                torch.cuda.synchronize()
                gi = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gi, stream=si):
                    yi = net(xi)
            flights.append((si, gi, xi, yi))
    torch.cuda.synchronize()
    hosts = [torch.randint(0, 256, (N, T, HW, HW, 3), dtype=torch.uint8, generator=g).pin_memory() for _ in range(nfl_max)]
    stages = [torch.empty((N, T, HW, HW, 3), dtype=torch.uint8, device=dev) for _ in range(nfl_max)]
    nbytes = hosts[0].numel()

    def loop(fn, secs=0.7):
        for k in range(8):
            fn(k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        k = 0
        while True:
            for _ in range(20):
                fn(k)
                k += 1
            if time.perf_counter() - t0 > secs:
                break
        torch.cuda.synchronize()
        return k / (time.perf_counter() - t0)

    out = {}
    for nfl in (1, 2, 3, 4)[:nfl_max]:
        def resident(k):
            fl = flights[k % nfl]
            with torch.cuda.stream(fl[0]):
                fl[1].replay()
        out["resident_%d" % nfl] = round(N * loop(resident), 1)
    for prio in (0, -1):
        cs = torch.cuda.Stream(priority=prio)
        for nfl in (2, 3, 4)[:max(0, nfl_max - 1)]:
            ev_c = [torch.cuda.Event() for _ in range(nfl)]
            ev_v = [torch.cuda.Event() for _ in range(nfl)]
            for i in range(nfl):
                ev_v[i].record(flights[i][0])

            def staged(k):
                i = k % nfl
                with torch.cuda.stream(cs):
                    cs.wait_event(ev_v[i])
                    stages[i].copy_(hosts[i], non_blocking=True)
                    ev_c[i].record(cs)
                with torch.cuda.stream(flights[i][0]):
                    flights[i][0].wait_event(ev_c[i])
                    ops.clip_from_u8(stages[i], scale=2, out=flights[i][2])
                    ev_v[i].record(flights[i][0])
                    flights[i][1].replay()
            out["staged_prio%d_%d" % (prio, nfl)] = round(N * loop(staged), 1)

        def copies(k):
            with torch.cuda.stream(cs):
                stages[k % 2].copy_(hosts[k % 2], non_blocking=True)
        out["copy_only_GBs_prio%d" % prio] = round(nbytes * loop(copies) / 1e9, 1)
    # flight 0 on the DEFAULT stream (what bench.py's resident loop does), high-priority copy stream
    cs = torch.cuda.Stream(priority=-1)
    dstreams = [torch.cuda.current_stream(), flights[1][0]]
    ev_c = [torch.cuda.Event() for _ in range(2)]
    ev_v = [torch.cuda.Event() for _ in range(2)]
    for i in range(2):
        ev_v[i].record(dstreams[i])

    def staged_default(k):
        i = k % 2
        with torch.cuda.stream(cs):
            cs.wait_event(ev_v[i])
            stages[i].copy_(hosts[i], non_blocking=True)
            ev_c[i].record(cs)
        with torch.cuda.stream(dstreams[i]):
            dstreams[i].wait_event(ev_c[i])
            ops.clip_from_u8(stages[i], scale=2, out=flights[i][2])
            ev_v[i].record(dstreams[i])
            flights[i][1].replay()
    out["staged_prio-1_2_flight0_on_default_stream"] = round(N * loop(staged_default), 1)
    # conversion on its own stream per flight (so the copy -> convert chain never sits in the compute stream's queue)
    for nfl in (2, 3)[:max(0, nfl_max - 1)]:
        cs = torch.cuda.Stream(priority=-1)
        ev_v = [torch.cuda.Event() for _ in range(nfl)]
        ev_g = [torch.cuda.Event() for _ in range(nfl)]
        for i in range(nfl):
            ev_g[i].record(flights[i][0])

        def staged2(k):
            i = k % nfl
            with torch.cuda.stream(cs):
                stages[i].copy_(hosts[i], non_blocking=True)
                cs.wait_event(ev_g[i])                           # the graph of step k - nfl has finished reading its input
                ops.clip_from_u8(stages[i], scale=2, out=flights[i][2])
                ev_v[i].record(cs)
            with torch.cuda.stream(flights[i][0]):
                flights[i][0].wait_event(ev_v[i])
                flights[i][1].replay()
                ev_g[i].record(flights[i][0])
        out["staged_convert_on_copy_stream_%d" % nfl] = round(N * loop(staged2), 1)
    try:
        for nfl in (2, 3)[:max(0, nfl_max - 1)]:
            def zero_copy(k):
                i = k % nfl
                with torch.cuda.stream(flights[i][0]):
                    ops.clip_from_u8(hosts[i], scale=2, out=flights[i][2])
                    flights[i][1].replay()
            out["zero_copy_%d" % nfl] = round(N * loop(zero_copy), 1)
        cs = torch.cuda.Stream(priority=-1)
        ev_v = [torch.cuda.Event() for _ in range(2)]
        ev_g = [torch.cuda.Event() for _ in range(2)]
        for i in range(2):
            ev_g[i].record(flights[i][0])

        def zero_copy_side(k):
            i = k % 2
            with torch.cuda.stream(cs):
                cs.wait_event(ev_g[i])
                ops.clip_from_u8(hosts[i], scale=2, out=flights[i][2])
                ev_v[i].record(cs)
            with torch.cuda.stream(flights[i][0]):
                flights[i][0].wait_event(ev_v[i])
                flights[i][1].replay()
                ev_g[i].record(flights[i][0])
        out["zero_copy_side_stream_2"] = round(N * loop(zero_copy_side), 1)

        def zc_only(k):
            ops.clip_from_u8(hosts[k % 2], scale=2, out=flights[k % 2][2])
        out["zero_copy_only_GBs"] = round(nbytes * loop(zc_only) / 1e9, 1)
    except Exception as e:
        out["zero_copy_error"] = repr(e)[:300]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
