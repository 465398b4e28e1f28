// Micro-benchmark: LDS fragment reads at 16-byte vs 4-byte alignment (gfx950).
//   mode 0: ds_read_b128, lane stride 16 B (what stem_tap_kernel does today)
//   mode 1: ds_read_b128 via asm, lane stride 12 B (4-byte aligned only)
//   mode 2: 2 x ds_read2_b32, lane stride 12 B
//   mode 3: ds_read_b128, lane stride 12 B rounded *down* to 16 B (control: same bank pattern class, aligned)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned* out, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[48 * 1024];
    for (int i = threadIdx.x; i < 12 * 1024; i += 256) ((unsigned*)lds)[i] = i * 2654435761u;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tw = lane & 15, th = (lane & 31) >> 4, kh = lane >> 5;
    const int stride = (MODE == 0) ? 16 : 12;
    unsigned addr = (unsigned)(size_t)lds + wave * 2048 + th * 320 + tw * stride + kh * 16;
    if (MODE == 3) addr &= ~15u;
    u32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            u32x4 v;
            if (MODE == 0 || MODE == 1 || MODE == 3) {
                asm volatile("ds_read_b128 %0, %1 offset:%2\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr), "n"(r * 640));
            } else {
                u32x2 a, b;
                asm volatile("ds_read2_b32 %0, %2 offset0:%3 offset1:%4\n ds_read2_b32 %1, %2 offset0:%5 offset1:%6\n s_waitcnt lgkmcnt(0)"
                             : "=v"(a), "=v"(b) : "v"(addr + r * 640), "n"(0), "n"(1), "n"(2), "n"(3));
                v = u32x4{a.x, a.y, b.x, b.y};
            }
            acc ^= v;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}

// pipelined variant: 8 reads in flight, one wait
template <int MODE>
__global__ __launch_bounds__(256) void kp(unsigned* out, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[48 * 1024];
    for (int i = threadIdx.x; i < 12 * 1024; i += 256) ((unsigned*)lds)[i] = i * 2654435761u;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tw = lane & 15, th = (lane & 31) >> 4, kh = lane >> 5;
    const int stride = (MODE == 0) ? 16 : 12;
    unsigned addr = (unsigned)(size_t)lds + wave * 2048 + th * 320 + tw * stride + kh * 16;
    u32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        u32x4 v[8];
        if (MODE != 2) {
            asm volatile(
                "ds_read_b128 %0, %8 offset:0\n ds_read_b128 %1, %8 offset:640\n ds_read_b128 %2, %8 offset:1280\n ds_read_b128 %3, %8 offset:1920\n"
                "ds_read_b128 %4, %8 offset:2560\n ds_read_b128 %5, %8 offset:3200\n ds_read_b128 %6, %8 offset:3840\n ds_read_b128 %7, %8 offset:4480\n"
                "s_waitcnt lgkmcnt(0)"
                : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]) : "v"(addr));
        } else {
            u32x2 a[16];
            asm volatile(
                "ds_read2_b32 %0, %16 offset0:0 offset1:1\n ds_read2_b32 %1, %16 offset0:2 offset1:3\n"
                "ds_read2_b32 %2, %16 offset0:160 offset1:161\n ds_read2_b32 %3, %16 offset0:162 offset1:163\n"
                "ds_read2_b32 %4, %17 offset0:0 offset1:1\n ds_read2_b32 %5, %17 offset0:2 offset1:3\n"
                "ds_read2_b32 %6, %17 offset0:160 offset1:161\n ds_read2_b32 %7, %17 offset0:162 offset1:163\n"
                "ds_read2_b32 %8, %18 offset0:0 offset1:1\n ds_read2_b32 %9, %18 offset0:2 offset1:3\n"
                "ds_read2_b32 %10, %18 offset0:160 offset1:161\n ds_read2_b32 %11, %18 offset0:162 offset1:163\n"
                "ds_read2_b32 %12, %19 offset0:0 offset1:1\n ds_read2_b32 %13, %19 offset0:2 offset1:3\n"
                "ds_read2_b32 %14, %19 offset0:160 offset1:161\n ds_read2_b32 %15, %19 offset0:162 offset1:163\n"
                "s_waitcnt lgkmcnt(0)"
                : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3]), "=&v"(a[4]), "=&v"(a[5]), "=&v"(a[6]), "=&v"(a[7]),
                  "=&v"(a[8]), "=&v"(a[9]), "=&v"(a[10]), "=&v"(a[11]), "=&v"(a[12]), "=&v"(a[13]), "=&v"(a[14]), "=&v"(a[15])
                : "v"(addr), "v"(addr + 1280), "v"(addr + 2560), "v"(addr + 3840));
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = u32x4{a[2 * r].x, a[2 * r].y, a[2 * r + 1].x, a[2 * r + 1].y};
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) acc ^= v[r];
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}

template <typename F>
static float timeit(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}

int main() {
    unsigned* out; hipMalloc(&out, 1024 * 256 * 4);
    const int iters = 4000, blocks = 1024;   // 4 blocks/CU resident (48 KB LDS each -> 3), 256 CUs
    const double bytes = (double)blocks * 256 * 16 * 8 * iters;
#define RUN(K, M) { float ms = timeit([&] { hipLaunchKernelGGL((K<M>), dim3(blocks), dim3(256), 0, 0, out, iters); }); \
        printf(#K " mode %d: %.3f ms  %.1f TB/s LDS  (%.1f B/clk/CU @2.4GHz)\n", M, ms, bytes / ms * 1e-9, bytes / (ms * 1e-3) / 256 / 2.4e9); }
    RUN(k, 0) RUN(k, 1) RUN(k, 2) RUN(k, 3)
    RUN(kp, 0) RUN(kp, 1) RUN(kp, 2)
    unsigned h[256]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    printf("check %u\n", h[5]);
    return 0;
}
