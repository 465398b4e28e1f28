// tools/ubench/winograd_kstep.hip -- what one K-step of a fused Winograd F(2x2,3x3)-over-HxW conv3d_2c would cost on gfx950, against
// the same skeleton of the shipped direct kernel ("prototype ... or a written no-go with SQ counters").
//
// This is an INSTRUCTION-MIX SKELETON, not a convolution: every workgroup runs, `iters` times, exactly the LDS / VALU / MFMA / global
// traffic one K-step of the design below needs (addresses are synthetic, results are garbage), one 512-thread workgroup per CU, and
// reports shader cycles (s_memtime) and wall time (s_memrealtime) per step.  It is OPTIMISTIC for Winograd: no index arithmetic, no
// halo predication, lane-linear (conflict-free) LDS addresses everywhere (SQ_LDS_BANK_CONFLICT = 0 in the counter pass), the transform's operands are already in registers after
// the raw reads, and the epilogue is charged separately from the numbers of MI355X_MICROARCH.md.
//
// Design W2 (the best one the register / LDS budgets admit): output tile 16x16 pixels of one plane = 64 Winograd tiles
// x 64 output channels per 8-wave workgroup (16 positions x 64 tiles x 64 channels fp32 = 128 accumulator registers per lane: more
// channels or tiles do not fit 256 VGPRs at two waves per SIMD); wave w owns positions 2w, 2w+1 (2x2 MFMA blocks each); K = 3 kd
// planes x 64 input channels = 12 steps of 16; per step and workgroup:
//      raw tile reads     64 tiles x 16 pixels x 32 B           = 32 KB   ds_read_b128
//      input transform    B^T d B per (tile, channel): 32 adds  (+ unpack / pack for bf16: 56 VALU ops per channel pair of 16 values)
//      V tile writes      16 positions x 64 tiles x 16 ch x 2 B = 32 KB   ds_write_b128
//      A fragments        2 positions x 2 tile blocks per wave  = 32 KB   ds_read_b128
//      B fragments        2 positions x 2 channel blocks per wave = 32 KB straight from L2 (the transformed weights of a position are
//                         used by ONE wave, so LDS staging would only add 32 KB of ds_write)
//      MFMA               8 per wave = 64 per workgroup (the direct kernel: 144 per two-tap step of its 256 x 192 tile)
//      barriers           2 (raw -> V, V -> fragments)
// MIX 0 = the direct kernel's step for comparison (per wave and tap: 5 fragment reads, 6 MFMAs, weights through LDS: 3 KB ds_write per wave
// and two taps), MIX 1 = W2 with a packed-fp16 transform (32 VALU ops), MIX 2 = W2 with the bf16 transform (unpack, fp32 adds, pack: 112).
//   hipcc --offload-arch=gfx950 -O2 winograd_kstep.hip -o winograd_kstep && ./winograd_kstep
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0)

template <int MIX>
__global__ __launch_bounds__(512) void k(unsigned long long* out, const s16x8* __restrict__ wts, int iters) {
    extern __shared__ s16x8 lds[];                              // 96 KB: raw 16 KB | V 2 x 32 KB | direct: halo + weights
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 6144; i += 512) { s16x8 v; for (int e = 0; e < 8; ++e) v[e] = (short)(0x3c00 + (i * 7 + e) % 977); lds[i] = v; }
    __syncthreads();
    f32x16 acc[8];
    for (int a = 0; a < 8; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), t0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (MIX == 0) {
            // direct two-tap step of the 256 x 192 tile, per wave: 2 x (2 A + 3 B) fragment reads, 12 MFMAs on 6 accumulators; weights of the
            // step (2 taps x 16 x 192 x 2 B = 12 KB per workgroup) global -> registers -> LDS
            s16x8 w0 = wts[((it * 2 + 0) * 512 + tid) & 65535], w1 = wts[((it * 2 + 1) * 512 + (tid & 255)) & 65535];       // 12 KB per workgroup
            s16x8 f[10];
#pragma unroll
            for (int q = 0; q < 10; ++q) f[q] = lds[(((it * 10 + q) * 64 + lane) * 5 + wave * 320) % 6144];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int a = 0; a < 6; ++a) acc[a] = MFMA(f[5 * t + (a & 1)], f[5 * t + 2 + a / 2], acc[a]);
            lds[(2048 + tid) % 6144] = w0;
            if (tid < 256) lds[(2560 + tid) % 6144] = w1;
            __syncthreads();
        } else {
            // B fragments of this wave's two positions straight from L2: 4 x 1 KB
            s16x8 b[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) b[q] = wts[((it * 8 + wave) * 4 + q) * 64 % 65536 + lane];
            // a thread transforms two (tile, 8-channel) items, one after the other (register budget: 128 accumulators + 16 B-fragment
            // registers leave ~100): column by column -- 4 raw pixels (ds_read_b128 each) -> 4 row-transformed values -- then row by row in
            // place, then 16 ds_write_b128 of the item's V values
            const int vb = 1024 + (it & 1) * 2048;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                s16x8 r[16];
                if (MIX == 1) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const s16x8 d0 = lds[(u * 512 + tid + (0 + c) * 64 + it) & 1023], d1 = lds[(u * 512 + tid + (4 + c) * 64 + it) & 1023];
                        const s16x8 d2 = lds[(u * 512 + tid + (8 + c) * 64 + it) & 1023], d3 = lds[(u * 512 + tid + (12 + c) * 64 + it) & 1023];
                        r[0 + c] = d0 - d2; r[4 + c] = d1 + d2; r[8 + c] = d2 - d1; r[12 + c] = d1 - d3;       // (v_pk_sub / v_pk_add on 16-bit pairs)
                    }
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const s16x8 a0 = r[4 * rr + 0], a1 = r[4 * rr + 1], a2 = r[4 * rr + 2], a3 = r[4 * rr + 3];
                        r[4 * rr + 0] = a0 - a2; r[4 * rr + 1] = a1 + a2; r[4 * rr + 2] = a2 - a1; r[4 * rr + 3] = a1 - a3;
                    }
                } else {
                    // bf16: unpack to fp32 (shift), fp32 adds, pack (here: truncation by shift; the real kernel would use v_cvt_pk_bf16_f32: half the ops)
                    float x[16][8];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const s16x8 d0 = lds[(u * 512 + tid + (0 + c) * 64 + it) & 1023], d1 = lds[(u * 512 + tid + (4 + c) * 64 + it) & 1023];
                        const s16x8 d2 = lds[(u * 512 + tid + (8 + c) * 64 + it) & 1023], d3 = lds[(u * 512 + tid + (12 + c) * 64 + it) & 1023];
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float f0 = __builtin_bit_cast(float, (unsigned)(unsigned short)d0[e] << 16), f1 = __builtin_bit_cast(float, (unsigned)(unsigned short)d1[e] << 16);
                            const float f2 = __builtin_bit_cast(float, (unsigned)(unsigned short)d2[e] << 16), f3 = __builtin_bit_cast(float, (unsigned)(unsigned short)d3[e] << 16);
                            x[0 + c][e] = f0 - f2; x[4 + c][e] = f1 + f2; x[8 + c][e] = f2 - f1; x[12 + c][e] = f1 - f3;
                        }
                    }
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float a0 = x[4 * rr + 0][e], a1 = x[4 * rr + 1][e], a2 = x[4 * rr + 2][e], a3 = x[4 * rr + 3][e];
                            r[4 * rr + 0][e] = (short)(__builtin_bit_cast(unsigned, a0 - a2) >> 16); r[4 * rr + 1][e] = (short)(__builtin_bit_cast(unsigned, a1 + a2) >> 16);
                            r[4 * rr + 2][e] = (short)(__builtin_bit_cast(unsigned, a2 - a1) >> 16); r[4 * rr + 3][e] = (short)(__builtin_bit_cast(unsigned, a1 - a3) >> 16);
                        }
                }
#pragma unroll
                for (int q = 0; q < 16; ++q) lds[vb + q * 128 + ((u * 512 + tid) & 127)] = r[q];
            }
            __syncthreads();
            // A fragments: 2 positions x 2 tile blocks
            s16x8 a[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) a[q] = lds[vb + ((wave * 2 + (q >> 1)) * 128 + (q & 1) * 64 + lane) % 2048];
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[4 * p + m] = MFMA(a[2 * p + (m & 1)], b[2 * p + (m >> 1)], acc[4 * p + m]);
            __syncthreads();
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), t1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int a = 0; a < 8; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    if (tid == 0) { out[2 * blockIdx.x] = c1 - c0; out[2 * blockIdx.x + 1] = t1 - t0; }
    if (s == 123.456f) out[0] = 0;
}

template <int MIX>
static void run(const char* name, int grid, int iters, double mfma_per_step, double direct_equiv_steps) {
    unsigned long long* d;
    s16x8* w;
    hipMalloc(&d, sizeof(unsigned long long) * 2 * grid);
    hipMalloc(&w, sizeof(s16x8) * 65536 + 4096);
    hipMemset(w, 0x3c, sizeof(s16x8) * 65536 + 4096);
    hipFuncSetAttribute((const void*)k<MIX>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    for (int rep = 0; rep < 3; ++rep) {
        k<MIX><<<grid, 512, 98304>>>(d, w, iters);
        hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h(2 * grid);
    hipMemcpy(h.data(), d, sizeof(unsigned long long) * 2 * grid, hipMemcpyDeviceToHost);
    std::vector<double> ghz, cyc, us;
    for (int i = 0; i < grid; ++i) if (h[2 * i + 1]) { ghz.push_back(h[2 * i] / (h[2 * i + 1] * 10.0)); cyc.push_back((double)h[2 * i] / iters); us.push_back(h[2 * i + 1] * 0.01 / iters); }
    std::sort(ghz.begin(), ghz.end()); std::sort(cyc.begin(), cyc.end()); std::sort(us.begin(), us.end());
    const double c = cyc[cyc.size() / 2], g = ghz[ghz.size() / 2], u = us[us.size() / 2];
    printf("%-34s grid %4d  clock %.2f GHz  cycles/step %7.0f  us/step %.4f  MFMA/step %4.0f  MFMA-only cycles %5.0f  (step / MFMA-only = %.2f)\n",
           name, grid, g, c, u, mfma_per_step, mfma_per_step / 4 * 32, c / (mfma_per_step / 4 * 32));
    (void)direct_equiv_steps;
    hipFree(d); hipFree(w);
}

int main() {
    int dev = 0; hipDeviceProp_t p; hipGetDeviceProperties(&p, dev);
    const int cus = p.multiProcessorCount;
    printf("%s, %d CUs; one 512-thread workgroup per CU, 2000 steps each\n", p.name, cus);
    run<0>("direct: two-tap step 256px x 192ch", cus, 2000, 96.0, 1.0);
    run<1>("winograd W2 step, packed-f16 transform", cus, 2000, 64.0, 1.0);
    run<2>("winograd W2 step, bf16 transform", cus, 2000, 64.0, 1.0);
    printf("\nper 256-pixel x 64-channel output block of conv3d_2c (Cin 64, K = 27 x 64 direct / 3 x 16 x 64 Winograd):\n"
           "  direct   : 1728 MFMAs = 18 two-tap steps of a third of the 192-channel tile (the step above covers 192 channels: 54 steps / 3)\n"
           "  winograd : 768 MFMAs = 12 steps above + the output transform (16 x 64 x 64 fp32 through LDS: 256 KB ds_write_b128 at 13 cycles / KB\n"
           "             + 256 KB ds_read_b128 at 4 cycles / KB = ~4350 cycles, MI355X_MICROARCH.md LDS table) + the raw-tile load\n");
    return 0;
}
