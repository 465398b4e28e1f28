// tests/emul/hipemu.h -- a tiny host-side SIMT interpreter for the kernels in step_amd/csrc.
//
// TEST INFRASTRUCTURE ONLY.  The build container has no GPU; this lets the SAME kernel source
// (compiled with -DSTEP_EMUL by clang++ for x86) run on the CPU so that indexing, tiling, LDS
// staging and MFMA fragment mapping can be checked against the oracle before GPU minutes are
// spent.  It says nothing about performance.  The product (libstep_amd.so) never contains or
// loads any of this.
//
// Model: a workgroup is a set of cooperative fibers (one per work-item) multiplexed on ONE OS
// thread; __syncthreads() and the wave-level exchange primitives yield to the scheduler.
// Different workgroups run on different OS threads; `__shared__` is `static thread_local`.
#pragma once
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <float.h>
#include <functional>

namespace hipemu {
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
extern thread_local dim3 t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
void launch(dim3 grid, dim3 block, const std::function<void()>& body);
void block_barrier();
// wave-level exchange: every lane of the calling wave deposits `bytes` (<=256) and can then
// read any lane's deposit.  exchange_begin() returns a pointer to the 64 x 256 B scratch AFTER
// all lanes have deposited; exchange_end() must be called by every lane before the next one.
char* exchange_begin(const void* mine, int bytes);
void exchange_end();
typedef float f32x16 __attribute__((ext_vector_type(16)));
void mfma_32x32_k16(const float (&a)[8], const float (&b)[8], f32x16& c, bool f32_pairing);
typedef float f32x4 __attribute__((ext_vector_type(4)));
// v_mfma_f32_16x16x32_{bf16,f16}: lane l holds row/col (l & 15), k = 8 * (l >> 4) + 0..7; D: col = l & 15, row = 4 * (l >> 4) + reg
void mfma_16x16_k32(const float (&a)[8], const float (&b)[8], f32x4& c);
// ds_read_b64_tr_b16: every lane supplies the 4 x 16-bit elements at its own address; lane (16 g + i) receives, as element j,
// element (i % 4) of the data of lane (16 g + 4 j + i / 4) (measured on gfx950, tools/ubench/tr_read.hip)
void tr16_b64(const unsigned short (&mine)[4], unsigned short (&out)[4]);
}  // namespace hipemu

using hipemu::dim3;
#define threadIdx (hipemu::t_threadIdx)
#define blockIdx (hipemu::t_blockIdx)
#define blockDim (hipemu::t_blockDim)
#define gridDim (hipemu::t_gridDim)
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static thread_local
#define __restrict__

typedef void* hipStream_t;

static inline void __syncthreads() { hipemu::block_barrier(); }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }

template <typename T> static inline T __shfl(T v, int src) {
    char* s = hipemu::exchange_begin(&v, (int)sizeof(T));
    T r;
    memcpy(&r, s + 256 * (src & 63), sizeof(T));
    hipemu::exchange_end();
    return r;
}
template <typename T> static inline T __shfl_xor(T v, int m) { return __shfl(v, (int)((threadIdx.x & 63) ^ m)); }
template <typename T> static inline T __shfl_down(T v, int d) {
    int l = (int)(threadIdx.x & 63);
    return __shfl(v, l + d < 64 ? l + d : l);
}
static inline unsigned long long __ballot(int pred) {
    int p = pred ? 1 : 0;
    char* s = hipemu::exchange_begin(&p, 4);
    unsigned long long m = 0;
    for (int i = 0; i < 64; ++i) {
        int q;
        memcpy(&q, s + 256 * i, 4);
        if (q) m |= 1ull << i;
    }
    hipemu::exchange_end();
    return m;
}
static inline int atomicMin(int* p, int v) {          // (LDS min / max of the kernels: the fibers of a block run on one OS thread, but stay atomic anyway)
    int old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (v < old && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}
static inline int atomicMax(int* p, int v) {
    int old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (v > old && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}
static inline float atomicAdd(float* p, float v) {
    // blocks may run on different OS threads
    unsigned int* ip = (unsigned int*)p;
    unsigned int old = __atomic_load_n(ip, __ATOMIC_RELAXED), nw;
    float f;
    do {
        memcpy(&f, &old, 4);
        f += v;
        memcpy(&nw, &f, 4);
    } while (!__atomic_compare_exchange_n(ip, &old, nw, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
    memcpy(&f, &old, 4);
    return f;
}
static inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }
static inline float __fsub_rn(float a, float b) { volatile float r = a - b; return r; }
static inline float __fmul_rn(float a, float b) { volatile float r = a * b; return r; }
static inline float __fdiv_rn(float a, float b) { volatile float r = a / b; return r; }
static inline int hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
