// tools/ubench/clock_vs_lds_ratio.hip -- does the LDS traffic of a matrix kernel cost CLOCK on MI355X (gfx950)?
// The conv kernels read 1.33 ds_read_b128 per v_mfma_f32_32x32x16_bf16 (one pixel fragment + three weight fragments per three
// MFMAs) and run at 1.3-1.5 GHz, bare MFMA at 1.86 GHz.  This kernel keeps the matrix pipe saturated (8 waves per CU, two per
// SIMD, two operand sets in registers: one is multiplied while the other is read) and varies ONLY the number of LDS fragment
// reads per four MFMAs (R4 = 0 .. 8, i.e. 0 .. 2 reads per MFMA), on pseudo-random operands.  It reports the clock each mix is
// granted (s_memtime / s_memrealtime) and the matrix rate -- the answer sizes the gain of larger register tiles (fewer LDS bytes
// per FLOP).
//   hipcc --offload-arch=gfx950 -O2 -Wno-unused-value clock_vs_lds_ratio.hip -o clock_vs_lds_ratio && ./clock_vs_lds_ratio
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ inline unsigned short rnd_bf16(unsigned h) {
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    return (unsigned short)((h & 0x807fu) | (0x3e80u + ((h >> 8) & 0x0180u)));      // +-[0.25, 2), random sign / mantissa
}

template <int R4>
__global__ __launch_bounds__(512) void k(unsigned long long* out, int iters) {
    __shared__ u16x8 lds[4096];                     // 64 KB
    for (int i = threadIdx.x; i < 4096; i += 512) {
        u16x8 v;
        for (int e = 0; e < 8; ++e) v[e] = rnd_bf16((i * 8u + e) * 2654435761u + blockIdx.x * 40503u);
        lds[i] = v;
    }
    __syncthreads();
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = (float)(a + 1);      // distinct: identical recurrences would be merged
    constexpr int NR = R4 ? R4 : 2;
    u16x8 s0[NR], s1[NR];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    auto rd = [&](u16x8* d, int kk) {
#pragma unroll
        for (int q = 0; q < NR; ++q) d[q] = lds[((kk * 37 + q * 5) * 512 + wave * 64 + lane) & 4095];      // a different 8 KB slice per read and step
    };
    auto mm = [&](const u16x8* c) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
            acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, c[(2 * a) % NR]), __builtin_bit_cast(bf16x8, c[(2 * a + 1) % NR]), acc[a], 0, 0, 0);
    };
    rd(s0, 0); rd(s1, 1);
    __syncthreads();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), t0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it += 2) {
        if (R4) rd(s1, it + 1);
        mm(s0);
        if (R4) rd(s0, it + 2);
        mm(s1);
    }
    __syncthreads();        // the WORKGROUP's time: the SIMD issues oldest-wave-first, wave 0 alone would report half of it
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), t1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = c1 - c0; out[2 * blockIdx.x + 1] = t1 - t0; }
    if (s == 123.456f) out[0] = 0;
}

template <int R4>
static void run(int grid, int iters) {
    unsigned long long* d;
    hipMalloc(&d, sizeof(unsigned long long) * 2 * grid);
    for (int w = 0; w < 30; ++w) k<R4><<<grid, 512>>>(d, iters);      // tens of ms of this load before the launch that is read
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(2 * grid);
    hipMemcpy(h.data(), d, sizeof(unsigned long long) * 2 * grid, hipMemcpyDeviceToHost);
    std::vector<double> ghz, us;
    for (int i = 0; i < grid; ++i) if (h[2 * i + 1]) { ghz.push_back(h[2 * i] / (h[2 * i + 1] * 10.0)); us.push_back(h[2 * i + 1] * 0.01); }
    std::sort(ghz.begin(), ghz.end()); std::sort(us.begin(), us.end());
    const double med = ghz[ghz.size() / 2], tmed = us[us.size() / 2];
    const double tf = grid * 8.0 * 4 * 32768 * iters / (tmed * 1e-6) / 1e12;       // 8 waves x 4 MFMAs x 32768 FLOP per step per workgroup
    const double cyc = tmed * 1e-6 * med * 1e9 / (2.0 * 4.0 * iters);              // two waves per SIMD: cycles of the pipe per MFMA
    printf("%d.%02d ds_read_b128 per MFMA: clock %.3f GHz (p5 %.3f, p95 %.3f)  %.1f cycles per MFMA per SIMD  %.0f TFLOP/s  LDS %.0f B/clk/CU\n",
           R4 / 4, (R4 % 4) * 25, med, ghz[ghz.size() / 20], ghz[ghz.size() - 1 - ghz.size() / 20], cyc, tf, R4 * 1024.0 / cyc);
    hipFree(d);
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int grid = p.multiProcessorCount, it = 6000;
    printf("# %d CUs, one 512-thread workgroup per CU, %d steps x 4 MFMAs per wave\n", grid, it);
    run<0>(grid, it); run<2>(grid, it); run<3>(grid, it); run<4>(grid, it); run<5>(grid, it); run<6>(grid, it); run<8>(grid, it);
    run<0>(grid, it);
    return 0;
}
