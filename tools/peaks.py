"""tools/peaks.py -- achievable peaks of the box next to the datasheet values: a device-to-device stream copy
(HBM read + write) and a large bf16 / fp32 GEMM through torch (hipBLASLt / rocBLAS) for the matrix cores."""
import torch

dev = torch.device("cuda:0")


def timed(fn, n):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


for mb in (256, 1024, 4096):
    x = torch.empty(mb << 20, dtype=torch.uint8, device=dev)
    y = torch.empty_like(x)
    t = timed(lambda: y.copy_(x), 20)
    print("stream copy %5d MiB: %7.1f GB/s (read + write)" % (mb, 2 * (mb << 20) / t / 1e9))
x = torch.zeros(1 << 30, dtype=torch.uint8, device=dev)
t = timed(lambda: x.zero_(), 20)
print("fill        1024 MiB: %7.1f GB/s (write only)" % ((1 << 30) / t / 1e9))
for dt, n in ((torch.bfloat16, 8192), (torch.float16, 8192), (torch.float32, 8192)):
    a = torch.randn(n, n, device=dev, dtype=dt)
    b = torch.randn(n, n, device=dev, dtype=dt)
    t = timed(lambda: torch.matmul(a, b), 10)
    print("GEMM %s %d^3: %7.1f TFLOP/s" % (str(dt).split(".")[-1], n, 2 * n ** 3 / t / 1e12))
