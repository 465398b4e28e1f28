// step_amd/csrc/diag.hip -- step_mfma_clock_probe: what this GPU sustains when every CU does nothing but 16-bit matrix
// instructions.  MI355X is power-managed: the datasheet's dense bf16 figure assumes 2.4 GHz, but with all 256 CUs issuing
// v_mfma_f32_32x32x16_bf16 back to back the boxes of this pool settle near 1.9 GHz.  bench.py reports the
// measured figure next to the datasheet roofline, so that a kernel's fraction can be read against what the box at hand can do.
// s_memtime counts shader cycles, s_memrealtime a constant 100 MHz; their ratio over the loop is the clock of that CU.
#include "common.h"

namespace step {

__global__ __launch_bounds__(256) void mfma_clock_probe_kernel(unsigned long long* __restrict__ out, int iters) {
    f32x16 acc[4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    u16x8 fa, fb;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        // pseudo-random bf16 operands in +-[0.25, 2): random sign and mantissa bits -- the power a matrix instruction draws depends on
        // how many operand bits toggle (near-constant operands measured 2.3 GHz on a box that grants random ones 1.9)
        unsigned h = (threadIdx.x * 8u + e) * 2654435761u + blockIdx.x * 40503u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        fa[e] = (unsigned short)((h & 0x807fu) | (0x3e80u + ((h >> 8) & 0x0180u)));
        fb[e] = (unsigned short)(((h >> 16) & 0x807fu) | (0x3e80u + ((h >> 24) & 0x0180u)));
    }
#ifdef STEP_EMUL
    const unsigned long long c0 = 0, t0 = 0;
#else
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), t0 = __builtin_amdgcn_s_memrealtime();
#endif
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int a = 0; a < 4; ++a) mma_k16(fa, fb, acc[a], bf16_t());      // four independent accumulators: the pipe never waits
    }
#ifdef STEP_EMUL
    const unsigned long long c1 = 0, t1 = 0;
#else
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), t1 = __builtin_amdgcn_s_memrealtime();
#endif
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    if (threadIdx.x == 0) {
        out[3 * (size_t)blockIdx.x + 0] = c1 - c0;
        out[3 * (size_t)blockIdx.x + 1] = t1 - t0;
        out[3 * (size_t)blockIdx.x + 2] = (unsigned long long)(s != 123.456f);      // (keeps the matrix work alive)
    }
}

// step_hbm_stream_probe: the plain streaming copy (16 B per lane, grid-stride, non-temporal) against which the HBM-bound kernels
// (pools, pointwise convs, ROIAlign) are read: what one launch moves per second on this box when it does nothing else.
__global__ __launch_bounds__(256) void hbm_stream_probe_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) {
#ifdef STEP_EMUL
        dst[i] = src[i];
#else
        __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
#endif
    }
}

// step_clock_sample: the EFFECTIVE shader clock while other work runs.  One wavefront per workgroup reads s_memtime (shader cycles) and
// s_memrealtime (100 MHz), sleeps until `ticks` of the 100 MHz counter have passed and reads both again; launched on a side stream beside
// a loop it costs one wave slot with a handful of registers and no memory traffic.  (The driver's DPM table, which sysfs / amd-smi report,
// is not this number: under matrix load the chip runs 1.3-1.9 GHz inside a 2.4 GHz DPM state.)
__global__ __launch_bounds__(64) void clock_sample_kernel(unsigned long long* __restrict__ out, unsigned ticks) {
#ifdef STEP_EMUL
    if (threadIdx.x == 0) { out[2 * (size_t)blockIdx.x] = 0; out[2 * (size_t)blockIdx.x + 1] = ticks; }
#else
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), t0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long t1 = t0;
    while (t1 - t0 < ticks) {
        __builtin_amdgcn_s_sleep(32);
        t1 = __builtin_amdgcn_s_memrealtime();
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[2 * (size_t)blockIdx.x] = c1 - c0; out[2 * (size_t)blockIdx.x + 1] = t1 - t0; }
#endif
}

}  // namespace step

using namespace step;

extern "C" {

int step_mfma_clock_probe(unsigned long long* out, int workgroups, int iters, step_stream_t stream) {
    if (workgroups < 0 || iters < 0) return STEP_E_SHAPE;
    if (workgroups == 0) return STEP_OK;
    if (!out) return STEP_E_NULL;
    STEP_LAUNCH(mfma_clock_probe_kernel, dim3((unsigned)workgroups), dim3(256), stream, out, iters);
    return STEP_LAUNCH_CHECK();
}

int step_clock_sample(unsigned long long* out, int workgroups, int ticks_100mhz, step_stream_t stream) {
    if (workgroups < 0 || ticks_100mhz < 0 || ticks_100mhz > 100000000) return STEP_E_SHAPE;
    if (workgroups == 0) return STEP_OK;
    if (!out) return STEP_E_NULL;
    STEP_LAUNCH(clock_sample_kernel, dim3((unsigned)workgroups), dim3(64), stream, out, (unsigned)ticks_100mhz);
    return STEP_LAUNCH_CHECK();
}

int step_hbm_stream_probe(const void* src, void* dst, size_t bytes, int workgroups, step_stream_t stream) {
    if (workgroups < 0 || (bytes & 15)) return STEP_E_SHAPE;
    if (bytes == 0 || workgroups == 0) return STEP_OK;
    if (!src || !dst) return STEP_E_NULL;
    if (((uintptr_t)src | (uintptr_t)dst) & 15) return STEP_E_ALIGN;
    STEP_LAUNCH(hbm_stream_probe_kernel, dim3((unsigned)workgroups), dim3(256), stream, (const u32x4*)src, (u32x4*)dst, bytes / 16);
    return STEP_LAUNCH_CHECK();
}

}  // extern "C"
