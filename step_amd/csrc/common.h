// step_amd/csrc/common.h -- shared device helpers for the gfx950 (CDNA4) kernels.
//
// One source, two builds:
//   * product : hipcc --offload-arch=gfx950  -> libstep_amd.so (the only thing step_amd loads)
//   * STEP_EMUL (tests/emul only): the same kernel bodies compiled for the host and run by a
//     fiber-based SIMT interpreter so that index logic can be checked without a GPU.  The
//     emulation build is test infrastructure; it is never loaded by the product path.
#pragma once
#include <stdint.h>
#include <float.h>
#include <math.h>

#ifdef STEP_EMUL
#include "hipemu.h"
#define STEP_WAVES_PER_SIMD(n)
#define STEP_WAVES_PER_SIMD_MIN(n)
#define STEP_SCHED_BARRIER()
#else
#include <hip/hip_runtime.h>
// register budget: make the compiler fit n wavefronts per SIMD (512 / n VGPRs each)
#define STEP_WAVES_PER_SIMD(n) __attribute__((amdgpu_waves_per_eu(n, n)))
#define STEP_WAVES_PER_SIMD_MIN(n) __attribute__((amdgpu_waves_per_eu(n)))
#define STEP_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)      // nothing is scheduled across this point
#endif

#include "../../include/step_amd.h"

namespace step {

typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// ---- storage element types ---------------------------------------------------------------
struct bf16_t { unsigned short v; };
struct f16_t { unsigned short v; };

__device__ __forceinline__ float bf16_bits_to_f32(unsigned short h) {
    unsigned int u = ((unsigned int)h) << 16;
    return __builtin_bit_cast(float, u);
}
// round-to-nearest-even, NaN kept quiet (same as torch's float -> bfloat16).  On the device this is the hardware
// conversion (v_cvt_pk_bf16_f32 on gfx950: one instruction per PAIR of values; the bit-twiddling form below costs ~7
// VALU operations per value and was a quarter of the epilogues' vector work); the host interpreter keeps the
// portable form.  Both are IEEE round-to-nearest-even, so finite results are bit-identical.
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) {
#ifndef STEP_EMUL
    return __builtin_bit_cast(unsigned short, (__bf16)f);
#endif
    unsigned int u = __builtin_bit_cast(unsigned int, f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float f16_bits_to_f32(unsigned short h) {
    _Float16 x = __builtin_bit_cast(_Float16, h);
    return (float)x;
}
__device__ __forceinline__ unsigned short f32_to_f16_bits(float f) {
    _Float16 x = (_Float16)f;
    return __builtin_bit_cast(unsigned short, x);
}

template <typename T> struct elem;
template <> struct elem<float> {
    static constexpr int dtype = STEP_F32;
    static constexpr int VEC = 4;  // elements per 16 bytes
    __device__ static __forceinline__ float to_f32(float x) { return x; }
    __device__ static __forceinline__ float from_f32(float x) { return x; }
    __device__ static __forceinline__ unsigned short bits16(float) { return 0; }      // 16-bit helpers: unused for fp32
    __device__ static __forceinline__ float from_bits16(unsigned short) { return 0.f; }
};
template <> struct elem<bf16_t> {
    static constexpr int dtype = STEP_BF16;
    static constexpr int VEC = 8;
    __device__ static __forceinline__ float to_f32(bf16_t x) { return bf16_bits_to_f32(x.v); }
    __device__ static __forceinline__ bf16_t from_f32(float x) { bf16_t r; r.v = f32_to_bf16_bits(x); return r; }
    __device__ static __forceinline__ unsigned short bits16(float x) { return f32_to_bf16_bits(x); }
    __device__ static __forceinline__ float from_bits16(unsigned short h) { return bf16_bits_to_f32(h); }
};
template <> struct elem<f16_t> {
    static constexpr int dtype = STEP_F16;
    static constexpr int VEC = 8;
    __device__ static __forceinline__ float to_f32(f16_t x) { return f16_bits_to_f32(x.v); }
    __device__ static __forceinline__ f16_t from_f32(float x) { f16_t r; r.v = f32_to_f16_bits(x); return r; }
    __device__ static __forceinline__ unsigned short bits16(float x) { return f32_to_f16_bits(x); }
    __device__ static __forceinline__ float from_bits16(unsigned short h) { return f16_bits_to_f32(h); }
};

// ---- 32x32 MFMA "k16 step": D(32x32) += A(32x16) * B(16x32) --------------------------------
// Operand convention used by every kernel here: lane l holds, for row/col (l & 31), the EIGHT
// consecutive k values  kbase + 8*(l>>5) + {0..7}  of its operand.
//   16-bit types: exactly the v_mfma_f32_32x32x16_{bf16,f16} fragment (one instruction).
//   fp32        : eight v_mfma_f32_32x32x2_f32, instruction j consuming element j of both
//                 operands (k = kbase + j from lanes 0-31 and kbase + 8 + j from lanes 32-63);
//                 exact fp32 FMA chain -- this is the parity path.
// C/D layout (dtype independent): col = lane & 31, row = (r & 3) + 8*(r >> 2) + 4*(lane >> 5).
template <typename T> struct frag;
template <> struct frag<float> { typedef f32x8 type; };
template <> struct frag<bf16_t> { typedef u16x8 type; };
template <> struct frag<f16_t> { typedef u16x8 type; };

#ifndef STEP_EMUL
typedef __bf16 bf16x8_hw __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_hw __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void mma_k16(const f32x8& a, const f32x8& b, f32x16& c, float) {
#pragma unroll
    for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], c, 0, 0, 0);
}
__device__ __forceinline__ void mma_k16(const u16x8& a, const u16x8& b, f32x16& c, bf16_t) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_hw, a), __builtin_bit_cast(bf16x8_hw, b), c, 0, 0, 0);
}
__device__ __forceinline__ void mma_k16(const u16x8& a, const u16x8& b, f32x16& c, f16_t) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_hw, a), __builtin_bit_cast(f16x8_hw, b), c, 0, 0, 0);
}
#else
// host interpreter versions (tests/emul/hipemu.h)
__device__ inline void mma_k16(const f32x8& a, const f32x8& b, f32x16& c, float) {
    float fa[8], fb[8];
    for (int j = 0; j < 8; ++j) { fa[j] = a[j]; fb[j] = b[j]; }
    hipemu::mfma_32x32_k16(fa, fb, c, /*f32 pairing*/ true);
}
__device__ inline void mma_k16(const u16x8& a, const u16x8& b, f32x16& c, bf16_t) {
    float fa[8], fb[8];
    for (int j = 0; j < 8; ++j) { fa[j] = bf16_bits_to_f32(a[j]); fb[j] = bf16_bits_to_f32(b[j]); }
    hipemu::mfma_32x32_k16(fa, fb, c, false);
}
__device__ inline void mma_k16(const u16x8& a, const u16x8& b, f32x16& c, f16_t) {
    float fa[8], fb[8];
    for (int j = 0; j < 8; ++j) { fa[j] = f16_bits_to_f32(a[j]); fb[j] = f16_bits_to_f32(b[j]); }
    hipemu::mfma_32x32_k16(fa, fb, c, false);
}
#endif

// ---- 16x16 MFMA "k32 step" + the LDS transpose read that feeds it (weight-gradient kernels: the reduction axis is the PIXEL
// axis of channels-last tensors).  D(16x16) += A(16x32) * B(32x16); lane l holds, for row / col (l & 15), the eight k values
// 8 * (l >> 4) + {0..7}.  C/D: col = l & 15, row = 4 * (l >> 4) + reg.
//   lds_tr8: ds_read_b64_tr_b16 x 2.  The 16 lanes of group g = l >> 4 each read 8 bytes of a PIXEL-major LDS image and
//   the hardware transposes the 4 x 16 block inside the group: lane 4 r + q supplies (pixel r of the block, channels
//   4 q .. 4 q + 3), lane i receives channel i of pixels 0..3 (measured, tools/ubench/tr_read.hip).  a0 / a1: THIS lane's
//   addresses for block pixels 0-3 and 4-7, i.e. the address of pixel (r = (l & 15) >> 2) of each half plus (l & 3) * 8 --
//   the pixels of a block need not be equidistant in LDS (a halo tile's rows are skewed against the output pixel order).
#ifndef STEP_EMUL
typedef short i16x4_hw __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u16x8 lds_tr8(const unsigned char* a0, const unsigned char* a1) {
    typedef __attribute__((address_space(3))) i16x4_hw* lptr;
    const i16x4_hw lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(const __attribute__((address_space(3))) unsigned char*)a0);
    const i16x4_hw hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(const __attribute__((address_space(3))) unsigned char*)a1);
    u16x8 r = {(unsigned short)lo[0], (unsigned short)lo[1], (unsigned short)lo[2], (unsigned short)lo[3],
               (unsigned short)hi[0], (unsigned short)hi[1], (unsigned short)hi[2], (unsigned short)hi[3]};
    return r;
}
__device__ __forceinline__ void mma16_k32(const u16x8& a, const u16x8& b, f32x4& c, bf16_t) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_hw, a), __builtin_bit_cast(bf16x8_hw, b), c, 0, 0, 0);
}
__device__ __forceinline__ void mma16_k32(const u16x8& a, const u16x8& b, f32x4& c, f16_t) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_hw, a), __builtin_bit_cast(f16x8_hw, b), c, 0, 0, 0);
}
#else
__device__ inline u16x8 lds_tr8(const unsigned char* a0, const unsigned char* a1) {
    u16x8 r;
    for (int h = 0; h < 2; ++h) {
        unsigned short mine[4], out[4];
        __builtin_memcpy(mine, h ? a1 : a0, 8);
        hipemu::tr16_b64(mine, out);
        for (int j = 0; j < 4; ++j) r[h * 4 + j] = out[j];
    }
    return r;
}
__device__ inline void mma16_k32(const u16x8& a, const u16x8& b, f32x4& c, bf16_t) {
    float fa[8], fb[8];
    for (int j = 0; j < 8; ++j) { fa[j] = bf16_bits_to_f32(a[j]); fb[j] = bf16_bits_to_f32(b[j]); }
    hipemu::mfma_16x16_k32(fa, fb, c);
}
__device__ inline void mma16_k32(const u16x8& a, const u16x8& b, f32x4& c, f16_t) {
    float fa[8], fb[8];
    for (int j = 0; j < 8; ++j) { fa[j] = f16_bits_to_f32(a[j]); fb[j] = f16_bits_to_f32(b[j]); }
    hipemu::mfma_16x16_k32(fa, fb, c);
}
#endif

// ---- LDS-DMA: each lane copies 16 B from its own global address to  lds_base + lane*16 ----------
// (global_load_lds_dwordx4: the LDS destination is the wave-uniform base + lane x 16, the data never
// passes through VGPRs; completion is tracked by vmcnt -- a following __syncthreads() drains it.)
#ifndef STEP_EMUL
__device__ __forceinline__ void glds16(const void* gsrc_lane, void* lds_base_uniform) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc_lane,
                                     (__attribute__((address_space(3))) void*)lds_base_uniform, 16, 0, 0);
}
#else
__device__ inline void glds16(const void* gsrc_lane, void* lds_base_uniform) {
    __builtin_memcpy((unsigned char*)lds_base_uniform + 16 * (threadIdx.x & 63), gsrc_lane, 16);
}
#endif

__device__ __forceinline__ int cd_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// v_permlane32_swap_b32 (gfx950): the UPPER 32 lanes of x trade places with the LOWER 32 lanes of y -- afterwards lane l < 32 holds
// (x[l], x[l + 32]) in (x, y) and lane l + 32 holds (y[l], y[l + 32]).  (lane semantics measured: tools/ubench/lane_swap.hip)
#ifndef STEP_EMUL
__device__ __forceinline__ void lane32_swap(unsigned& x, unsigned& y) {
    const auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    x = r[0]; y = r[1];
}
#else
__device__ inline void lane32_swap(unsigned& x, unsigned& y) {
    const unsigned mine[2] = {x, y};
    const int lane = (int)(threadIdx.x & 63);
    char* s = hipemu::exchange_begin(mine, 8);
    unsigned nx = x, ny = y;
    if (lane >= 32) __builtin_memcpy(&nx, s + 256 * (lane - 32) + 4, 4);      // x.upper <- y.lower
    else __builtin_memcpy(&ny, s + 256 * (lane + 32), 4);                      // y.lower <- x.upper
    hipemu::exchange_end();
    x = nx; y = ny;
}
#endif

// ---- launch helper -------------------------------------------------------------------------
#ifdef STEP_EMUL
#define STEP_LAUNCH(kernel, grid, block, stream, ...) \
    hipemu::launch((grid), (block), [=]() { kernel(__VA_ARGS__); })
#define STEP_LAUNCH_CHECK() 0
#else
#define STEP_LAUNCH(kernel, grid, block, stream, ...) \
    hipLaunchKernelGGL(kernel, (grid), (block), 0, (hipStream_t)(stream), __VA_ARGS__)
#define STEP_LAUNCH_CHECK() ((int)hipGetLastError())
#endif

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline long long ceil_div64(long long a, long long b) { return (a + b - 1) / b; }

}  // namespace step
