// tools/ubench/clock_under_load.hip -- the shader clock an MI355X really runs at under different instruction mixes (gfx950).
// s_memtime counts shader cycles, s_memrealtime a fixed 100 MHz: their ratio over a kernel's life is the clock of that CU.
// Every workgroup runs `iters` rounds of its mix and reports cycles and time; the host prints median clock and the achieved
// MFMA rate.  Mixes: 0 = dependent v_fma chain (light), 1 = v_mfma_f32_32x32x16_bf16 back to back (4 accumulators per wave),
// 2 = MFMA + ds_read_b128 (6 reads per 4 MFMAs, the ratio of the conv kernels).
//   hipcc --offload-arch=gfx950 -O2 clock_under_load.hip -o clock_under_load && ./clock_under_load
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MIX>
__global__ __launch_bounds__(256) void k(unsigned long long* out, int iters) {
    __shared__ s16x8 lds[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) { s16x8 v; for (int e = 0; e < 8; ++e) v[e] = (short)(0x3c00 + i + e); lds[i] = v; }
    __syncthreads();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), t0 = __builtin_amdgcn_s_memrealtime();
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float x = threadIdx.x * 1e-3f;
    s16x8 fa = lds[threadIdx.x], fb = lds[256 + threadIdx.x];
    for (int it = 0; it < iters; ++it) {
        if (MIX == 0) {
#pragma unroll
            for (int u = 0; u < 32; ++u) x = x * 1.0001f + 0.5f;
        } else {
            if (MIX == 2) {
                s16x8 t[6];
#pragma unroll
                for (int q = 0; q < 6; ++q) t[q] = lds[((it * 6 + q) * 64 + threadIdx.x) & 2047];
                fa = t[0] ^ t[2] ^ t[4]; fb = t[1] ^ t[3] ^ t[5];
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa), __builtin_bit_cast(bf16x8, fb), acc[a], 0, 0, 0);
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), t1 = __builtin_amdgcn_s_memrealtime();
    float s = x;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = c1 - c0; out[2 * blockIdx.x + 1] = t1 - t0; }
    if (s == 123.456f) out[0] = 0;
}

template <int MIX>
static void run(const char* name, int grid, int iters) {
    unsigned long long* d;
    hipMalloc(&d, sizeof(unsigned long long) * 2 * grid);
    k<MIX><<<grid, 256>>>(d, iters);           // warm
    hipDeviceSynchronize();
    k<MIX><<<grid, 256>>>(d, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(2 * grid);
    hipMemcpy(h.data(), d, sizeof(unsigned long long) * 2 * grid, hipMemcpyDeviceToHost);
    std::vector<double> ghz, us;
    for (int i = 0; i < grid; ++i) if (h[2 * i + 1]) { ghz.push_back(h[2 * i] / (h[2 * i + 1] * 10.0)); us.push_back(h[2 * i + 1] * 0.01); }
    std::sort(ghz.begin(), ghz.end()); std::sort(us.begin(), us.end());
    const double med = ghz[ghz.size() / 2], tmed = us[us.size() / 2];
    // MFMA rate of one workgroup (4 waves, one per SIMD when alone on the CU): 4 waves x 4 MFMAs x 32768 flop per iteration
    const double tf_cu = MIX ? 4.0 * 4 * 32768 * iters / (tmed * 1e-6) / 1e12 : 0.0;
    printf("%-26s grid %5d: clock median %.2f GHz (p5 %.2f, p95 %.2f), %.1f us per workgroup%s", name, grid, med, ghz[ghz.size() / 20],
           ghz[ghz.size() - 1 - ghz.size() / 20], tmed, MIX ? "" : "\n");
    if (MIX) printf(", %.2f TFLOP/s per workgroup = %.0f cycles per MFMA per SIMD\n", tf_cu, tmed * 1e-6 * med * 1e9 / (4.0 * iters));
    hipFree(d);
}

int main() {
    const int it = 20000;
    run<0>("v_fma chain", 1, it); run<0>("v_fma chain", 256, it); run<0>("v_fma chain", 1024, it);
    run<1>("mfma 32x32x16 bf16", 1, it); run<1>("mfma 32x32x16 bf16", 256, it); run<1>("mfma 32x32x16 bf16", 768, it); run<1>("mfma 32x32x16 bf16", 2048, it);
    run<2>("mfma + 6 ds_read_b128 / 4", 256, it); run<2>("mfma + 6 ds_read_b128 / 4", 768, it); run<2>("mfma + 6 ds_read_b128 / 4", 2048, it);
    return 0;
}
