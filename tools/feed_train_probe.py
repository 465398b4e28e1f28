"""tools/feed_train_probe.py -- where the fed training loop of train_step_amd.py loses its time (GPU only, tuning aid; round 6: 40.5 ms per fed
iteration against 13.4 ms resident).  The captured C4 step with (a) nothing, (b) the conversion pass from a resident staging buffer, (c) the
host -> device copy on the MAIN stream + conversion, (d) train_step_amd.py's arrangement: copy on a high-priority copy stream, (e) the same on a
default-priority copy stream, (f) the copy alone."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from step_amd import ops, workloads  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    w = workloads.C4TrainStep(dev, batch=1, tubes_per_clip=5, seed=123, dtype=torch.bfloat16, capturable=True)
    w.capture(warmup=3)
    N, T, _, H, W = w.x.shape
    host = [torch.randint(0, 256, (N, T, H, W, 3), dtype=torch.uint8).pin_memory() for _ in range(2)]
    stage = [torch.empty((N, T, H, W, 3), dtype=torch.uint8, device=dev) for _ in range(2)]
    main = torch.cuda.current_stream()

    def timed(fn, iters=30):
        for k in range(3):
            fn(k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(iters):
            fn(k)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters * 1e3

    def a_(k):
        w.step()

    def b_(k):
        ops.clip_from_u8(stage[k % 2], scale=2, out=w.x)
        w.step()

    def c_(k):
        stage[k % 2].copy_(host[k % 2], non_blocking=True)
        ops.clip_from_u8(stage[k % 2], scale=2, out=w.x)
        w.step()

    def mk(prio):
        cs = torch.cuda.Stream(priority=prio)
        ev = [torch.cuda.Event() for _ in range(2)]

        def pre(k):
            with torch.cuda.stream(cs):
                stage[k % 2].copy_(host[k % 2], non_blocking=True)
                ev[k % 2].record(cs)

        def fn(k):
            if k == 0:
                pre(0)
            main.wait_event(ev[k % 2])
            ops.clip_from_u8(stage[k % 2], scale=2, out=w.x)
            cs.wait_stream(main)
            pre(k + 1)
            w.step()
        return fn

    def f_(k):
        stage[k % 2].copy_(host[k % 2], non_blocking=True)
    print("(a) captured step alone            %.3f ms" % timed(a_))
    print("(b) + conversion from resident u8  %.3f ms" % timed(b_))
    print("(c) + copy on the main stream      %.3f ms" % timed(c_))
    print("(d) copy on a priority -1 stream   %.3f ms" % timed(mk(-1)))
    print("(e) copy on a priority 0 stream    %.3f ms" % timed(mk(0)))
    print("(f) the copy alone                 %.3f ms (%.1f GB/s)" % ((lambda t: (t, host[0].numel() / t / 1e6))(timed(f_))))


if __name__ == "__main__":
    main()
