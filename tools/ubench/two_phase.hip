// tools/ubench/two_phase.hip -- why do the 24 back-to-back MFMAs of conv_tap's multiply phase take 0.56 us (~43 cycles each at the ~1.85 GHz the
// counters report) instead of 32 cycles each?  (round 6)
// The skeleton of the two-phase K loop (conv_tap_kernel.h, PH = 1): 8 waves, two groups in anti-phase -- while the waves of one group issue 24
// v_mfma_f32_32x32x16_bf16 out of registers, their SIMD partners of the other group read the next step's 20 fragments from LDS (ds_read_b128);
// barrier; roles swap.  Wave 0 stamps s_memtime around its multiply phases.  Variants:
//   READS = 0 / 10 / 20 / 30   fragment reads of the load phase (0: the partner only waits at the barrier)
//   AGPR  = 0 / 1              accumulators in VGPRs ("+v") or in the accumulator half of the register file ("+a"): LDS returns are written into
//                              VGPRs while the partner's MFMAs read and write their 16-register accumulators -- do they share ports?
//   hipcc --offload-arch=gfx950 -O2 two_phase.hip -o two_phase && ./two_phase
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
    return (unsigned short)((h & 0x807fu) | (0x3e80u + ((h >> 8) & 0x0180u)));
}

template <bool AGPR>
__device__ __forceinline__ void mfma(f32x16& acc, const u16x8& a, const u16x8& b) {
    if constexpr (AGPR) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}

template <int READS, bool AGPR, int NM = 24>
__global__ __launch_bounds__(512, 2) void k(unsigned long long* out, int iters) {
    __shared__ u16x8 lds[8192];                     // 128 KB
    for (int i = threadIdx.x; i < 8192; i += 512) {
        u16x8 v;
        for (int e = 0; e < 8; ++e) v[e] = rnd_bf16((i * 8u + e) * 2654435761u + blockIdx.x * 40503u);
        lds[i] = v;
    }
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wave >> 2;                      // waves w and w + 4 share a SIMD
    f32x16 acc[6];
    for (int a = 0; a < 6; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = (float)(a + 1);
    constexpr int NF = NM == 24 ? 20 : 30;         // fragments of a step: 2 taps x 2 k16 x (2 + 3), or three taps
    u16x8 f[NF];
    for (int q = 0; q < NF; ++q) f[q] = lds[(q * 512 + wave * 64 + lane) & 8191];
    __syncthreads();
    unsigned long long cphase = 0;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), t0 = __builtin_amdgcn_s_memrealtime();
    if (grp == 1) __builtin_amdgcn_s_barrier();     // anti-phase
    for (int it = 0; it < iters; ++it) {
        // ---- L: this step's fragments
#pragma unroll
        for (int q = 0; q < READS; ++q) f[q % NF] = lds[((it * 37 + q * 5) * 512 + wave * 64 + lane) & 8191];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        // ---- C: 24 MFMAs out of registers
        const unsigned long long m0 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < NM / 6; ++j)
#pragma unroll
            for (int a = 0; a < 6; ++a) mfma<AGPR>(acc[a], f[(j * 5 + a % 3) % NF], f[(j * 5 + 3 + a / 3) % NF]);
        __builtin_amdgcn_s_setprio(0);
        const unsigned long long m1 = __builtin_amdgcn_s_memtime();
        cphase += m1 - m0;
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), t1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int a = 0; a < 6; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    if (threadIdx.x == 0) { out[3 * blockIdx.x] = c1 - c0; out[3 * blockIdx.x + 1] = t1 - t0; out[3 * blockIdx.x + 2] = cphase; }
    if (s == 123.456f) out[0] = 0;
}

template <int READS, bool AGPR, int NM = 24>
static void run(int grid, int iters) {
    unsigned long long* d;
    hipMalloc(&d, sizeof(unsigned long long) * 3 * grid);
    for (int w = 0; w < 20; ++w) k<READS, AGPR, NM><<<grid, 512>>>(d, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(3 * grid);
    hipMemcpy(h.data(), d, sizeof(unsigned long long) * 3 * grid, hipMemcpyDeviceToHost);
    std::vector<double> ghz, half, cph, us;
    for (int i = 0; i < grid; ++i) if (h[3 * i + 1]) {
        ghz.push_back(h[3 * i] / (h[3 * i + 1] * 10.0));
        half.push_back((double)h[3 * i] / (2.0 * iters));          // cycles per half-step (one multiply phase of either group)
        cph.push_back((double)h[3 * i + 2] / iters);               // cycles of wave 0's multiply phase (issue of its 24 MFMAs)
        us.push_back(h[3 * i + 1] * 0.01);
    }
    auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    const double g = med(ghz), hs = med(half), cp = med(cph), t = med(us);
    printf("reads %2d  acc in %s: clock %.3f GHz | half-step %.0f cycles = %.3f us | wave 0's %d MFMAs issue in %.0f cycles (%.1f per MFMA) | %.0f TFLOP/s\n",
           READS, AGPR ? "AGPRs" : "VGPRs", g, hs, hs / g * 1e-3, NM, cp, cp / NM, grid * 8.0 * NM * 32768.0 * iters / (t * 1e-6) / 1e12);
    hipFree(d);
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int grid = p.multiProcessorCount, it = 4000;
    printf("# %d CUs, one 512-thread workgroup per CU, %d steps of (load phase | barrier | 24 MFMAs | barrier), two wave groups in anti-phase\n", grid, it);
    run<0, false>(grid, it); run<10, false>(grid, it); run<20, false>(grid, it); run<30, false>(grid, it);
    run<0, true>(grid, it); run<10, true>(grid, it); run<20, true>(grid, it); run<30, true>(grid, it);
    run<20, false>(grid, it);
    // three taps per phase: 36 MFMAs out of 30 fragments between two phase switches
    run<30, false, 36>(grid, it); run<20, false>(grid, it); run<30, false, 36>(grid, it);
    return 0;
}
