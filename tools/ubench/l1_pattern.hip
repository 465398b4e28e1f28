// tools/ubench/l1_pattern.hip -- does the vector L1 care how many cache lines one wave-wide 16-byte load touches?  (round 6)
// conv_pws_kernel loads its activations straight into the MFMA B-operand layout: lane = pixel (32 rows), two 8-channel groups -> one load
// instruction covers 32 rows x 32 bytes = 32 different 128-byte lines, and each line is touched by 4 consecutive instructions.  A row-major
// copy covers 8 rows x 128 bytes = 8 lines per instruction.  Same bytes, same buffer (L2 / MALL resident), same loads in flight:
//   P32: lane l -> row l % 32, bytes [32 j + 16 (l / 32), +16)        P8: lane l -> row 4 (j % 4) * 0 + ... see code
//   hipcc --offload-arch=gfx950 -O2 l1_pattern.hip -o l1_pattern && ./l1_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int PAT, int NL>
__global__ __launch_bounds__(512) void k(const unsigned char* __restrict__ x, unsigned* out, long long rows, int pitch, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long ngroups = rows / 32;
    long long g = (long long)blockIdx.x * 8 + wave;
    u32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        for (long long gg = g; gg < ngroups; gg += (long long)gridDim.x * 8) {
            u32x4 v[NL];
#pragma unroll
            for (int j = 0; j < NL; ++j) {
                size_t off;
                if (PAT == 32) off = (size_t)(gg * 32 + (lane & 31)) * pitch + j * 32 + 16 * (lane >> 5);
                else if (PAT == 16) off = (size_t)(gg * 32 + (j % 2) * 16 + (lane >> 2)) * pitch + (j / 2) * 64 + 16 * (lane & 3);   // conv_pw_kernel's slab staging: 16 rows x 64 B
                else off = (size_t)(gg * 32 + (j % 4) * 8 + (lane >> 3)) * pitch + (j / 4) * 128 + 16 * (lane & 7);
                v[j] = *(const u32x4*)(x + off);
            }
#pragma unroll
            for (int j = 0; j < NL; ++j) acc ^= v[j];
        }
    }
    if (acc[0] == 0x12345u && acc[1] == 7u) out[0] = acc[2] + acc[3];
}

template <int PAT, int NL>
static void run(const unsigned char* x, unsigned* out, long long rows, int pitch, const char* what) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 20;
    k<PAT, NL><<<256, 512>>>(x, out, rows, pitch, 2);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<PAT, NL><<<256, 512>>>(x, out, rows, pitch, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)rows * NL * 32 * iters;
    printf("%-44s rows %lld pitch %d: %.1f us per pass, %.2f TB/s\n", what, rows, pitch, ms * 1e3 / iters, bytes / (ms * 1e-3) / 1e12);
}

int main() {
    const long long rows = 59976 / 32 * 32;
    unsigned char* x; unsigned* out;
    hipMalloc(&x, (size_t)rows * 2048 + 4096); hipMemset(x, 1, (size_t)rows * 2048 + 4096); hipMalloc(&out, 64);
    // 256 channels of bf16 = 512 bytes per row = 16 loads of 32 bytes per row (the pws kernel's S = 4: 16 fragments)
    run<32, 16>(x, out, rows, 512, "lane = row, 32 B per row per load (pws)");
    run<8, 16>(x, out, rows, 512, "8 rows x 128 B per load (row-major)");
    run<32, 16>(x, out, rows, 2048, "lane = row, 32 B (pitch 2048: a slice of 1024 ch)");
    run<8, 16>(x, out, rows, 2048, "8 rows x 128 B (pitch 2048)");
    run<16, 16>(x, out, rows, 512, "16 rows x 64 B per load (conv_pw slab staging)");
    run<16, 16>(x, out, rows, 2048, "16 rows x 64 B (pitch 2048)");
    run<16, 64>(x, out, rows, 2048, "16 rows x 64 B, whole 2048-B rows, 64 loads");
    run<32, 64>(x, out, rows, 2048, "lane = row, whole 2048-B rows, 64 loads");
    run<8, 64>(x, out, rows, 2048, "8 rows x 128 B, whole 2048-B rows, 64 loads");
    return 0;
}
