"""tools/nms_call_latency.py -- host latency of one `step_amd.roi_layers.nms` call the way test.py:158-195 makes them (CPU tensors,
a few dozen boxes, 180 x B calls per batch): the call is an upload, one launch and a download; this prints microseconds per call."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from step_amd.roi_layers import nms  # noqa: E402


def main():
    torch.manual_seed(0)
    for k in (8, 40, 200):
        c = torch.rand(k, 2) * 200
        boxes = torch.cat([c, c + torch.rand(k, 2) * 80 + 4], 1)
        scores = torch.rand(k)
        for place in ("cpu", "cuda"):
            b, s = boxes.to(place), scores.to(place)
            for _ in range(20):
                nms(b, s, 0.4)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 300
            for _ in range(n):
                nms(b, s, 0.4)
            torch.cuda.synchronize()
            print("k = %3d, %4s tensors: %.1f us per call" % (k, place, (time.perf_counter() - t0) / n * 1e6))


if __name__ == "__main__":
    main()
