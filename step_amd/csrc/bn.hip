// step_amd/csrc/bn.hip -- batch-statistics BatchNorm3d (+ ReLU) of a conv unit, forward and backward.
//
// The reference runs its BatchNorm layers in eval mode whenever --freeze_stats is on (every shipped script; models/networks.py:85-99)
// and those fold into the conv epilogue (conv_*.hip).  With --freeze_stats False the layers stay in TRAINING mode
// (models/i3dpt.py:95-110: nn.BatchNorm3d after the conv; models/two_branch.py:160,372 for the heads' Inception blocks): every
// channel is normalised with the mean / biased variance of THIS batch over (N, D, H, W), the running statistics move by
// `momentum` towards the batch mean / UNBIASED variance, and backward flows through the statistics.
//
//   forward   z [M pixels, C] channels-last (the raw conv output)  ->  y = relu?(gamma * (z - mean) * invstd + beta)
//             + save_mean / save_invstd (fp32 [C], what backward needs) + the running-statistics update
//   backward  g = gy * (y > 0)?;  gbeta = sum g;  ggamma = sum g * xhat;  gz = gamma * invstd * (g - gbeta / M - xhat * ggamma / M)
//
// Statistics: a workgroup reduces a chunk of pixels per channel as SHIFTED sums (shift = the chunk's first pixel: no cancellation
// when |mean| >> std), leaves (n, mean, M2) per channel in a caller-owned workspace, and one finishing thread per channel merges
// the chunks in chunk order in double precision (Chan's update): deterministic, no atomics.  HBM-bound: z is read twice on the
// forward (statistics, then normalise) and z, y, gy twice on the backward.
#include "common.h"

namespace step {

constexpr int BN_CHUNK = 2048;         // pixels per workgroup of the reduction passes
constexpr int BN_V = 4;                // channels per lane

template <typename T>
__device__ __forceinline__ void bn_load4(const T* p, float (&v)[BN_V]);
template <>
__device__ __forceinline__ void bn_load4<float>(const float* p, float (&v)[BN_V]) {
    const f32x4 r = *(const f32x4*)p;
    v[0] = r[0]; v[1] = r[1]; v[2] = r[2]; v[3] = r[3];
}
template <>
__device__ __forceinline__ void bn_load4<bf16_t>(const bf16_t* p, float (&v)[BN_V]) {
    const u16x4 r = *(const u16x4*)p;
#pragma unroll
    for (int i = 0; i < BN_V; ++i) v[i] = bf16_bits_to_f32(r[i]);
}
template <>
__device__ __forceinline__ void bn_load4<f16_t>(const f16_t* p, float (&v)[BN_V]) {
    const u16x4 r = *(const u16x4*)p;
#pragma unroll
    for (int i = 0; i < BN_V; ++i) v[i] = f16_bits_to_f32(r[i]);
}
template <typename T>
__device__ __forceinline__ void bn_store4(T* p, const float (&v)[BN_V]);
template <>
__device__ __forceinline__ void bn_store4<float>(float* p, const float (&v)[BN_V]) {
    *(f32x4*)p = f32x4{v[0], v[1], v[2], v[3]};
}
template <>
__device__ __forceinline__ void bn_store4<bf16_t>(bf16_t* p, const float (&v)[BN_V]) {
    *(u16x4*)p = u16x4{f32_to_bf16_bits(v[0]), f32_to_bf16_bits(v[1]), f32_to_bf16_bits(v[2]), f32_to_bf16_bits(v[3])};
}
template <>
__device__ __forceinline__ void bn_store4<f16_t>(f16_t* p, const float (&v)[BN_V]) {
    *(u16x4*)p = u16x4{f32_to_f16_bits(v[0]), f32_to_f16_bits(v[1]), f32_to_f16_bits(v[2]), f32_to_f16_bits(v[3])};
}

// thread layout of the reduction passes: cg = channel group (4 channels) of this thread, pr = its pixel row, rows = pixel rows per
// workgroup; C / 4 <= 256 channel groups per pass (wider layers loop over passes of 256 groups)
struct BnLayout { int cgs, rows; };
__host__ __device__ inline BnLayout bn_layout(int C) {
    BnLayout l;
    const int g = C / BN_V;
    l.cgs = g < 256 ? g : 256;
    l.rows = 256 / l.cgs;
    return l;
}

// partial statistics of chunk blockIdx.x: part[(chunk * C + c) * 2 + {0, 1}] = local mean, M2 (n follows from the chunk index)
template <typename T>
__global__ __launch_bounds__(256) void bn_stats_kernel(const T* __restrict__ z, int zcs, long long M, int C, float* __restrict__ part) {
    __shared__ float red[2][256][BN_V];
    const long long p0 = (long long)blockIdx.x * BN_CHUNK;
    const int n = (int)((M - p0) < BN_CHUNK ? (M - p0) : BN_CHUNK);
    const BnLayout L = bn_layout(C);
    const int tid = threadIdx.x, cg = tid % L.cgs, pr = tid / L.cgs;
    for (int c0 = 0; c0 < C; c0 += L.cgs * BN_V) {
        const int c = c0 + cg * BN_V;
        const bool live = pr < L.rows && c < C;
        float K[BN_V] = {0.f, 0.f, 0.f, 0.f}, s[BN_V] = {0.f, 0.f, 0.f, 0.f}, ss[BN_V] = {0.f, 0.f, 0.f, 0.f};
        if (live) {
            bn_load4<T>(z + (size_t)p0 * zcs + c, K);
            for (int i = pr; i < n; i += L.rows) {
                float v[BN_V];
                bn_load4<T>(z + (size_t)(p0 + i) * zcs + c, v);
#pragma unroll
                for (int e = 0; e < BN_V; ++e) { const float d = v[e] - K[e]; s[e] += d; ss[e] += d * d; }
            }
        }
#pragma unroll
        for (int e = 0; e < BN_V; ++e) { red[0][tid][e] = s[e]; red[1][tid][e] = ss[e]; }
        __syncthreads();
        if (pr == 0 && c < C) {
#pragma unroll
            for (int e = 0; e < BN_V; ++e) {
                float S = 0.f, SS = 0.f;
                for (int r = 0; r < L.rows; ++r) { S += red[0][r * L.cgs + cg][e]; SS += red[1][r * L.cgs + cg][e]; }      // fixed order
                const float mean_d = S / (float)n;
                part[((size_t)blockIdx.x * C + c + e) * 2 + 0] = K[e] + mean_d;
                part[((size_t)blockIdx.x * C + c + e) * 2 + 1] = SS - S * mean_d;
            }
        }
        __syncthreads();
    }
}

// one thread per channel: Chan's merge of the chunk statistics in chunk order (double), then everything per-channel the other
// kernels need: save_mean, save_invstd, the folded scale / shift of the normalise pass, the running statistics.
__global__ void bn_finish_kernel(const float* __restrict__ part, int chunks, long long M, int C, const float* __restrict__ gamma,
                                 const float* __restrict__ beta, float eps, float momentum, float* __restrict__ running_mean,
                                 float* __restrict__ running_var, float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                 float* __restrict__ scale, float* __restrict__ shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double n = 0.0, mean = 0.0, m2 = 0.0;
    for (int k = 0; k < chunks; ++k) {
        const long long p0 = (long long)k * BN_CHUNK;
        const double nb = (double)((M - p0) < BN_CHUNK ? (M - p0) : BN_CHUNK);
        const double mb = (double)part[((size_t)k * C + c) * 2 + 0], m2b = (double)part[((size_t)k * C + c) * 2 + 1];
        const double tot = n + nb, delta = mb - mean;
        mean = mean + delta * (nb / tot);
        m2 = m2 + m2b + delta * delta * (n * nb / tot);
        n = tot;
    }
    const double var = m2 / n;                                     // biased: what the batch is normalised with
    const float fmean = (float)mean, fvar = (float)var;
    const float invstd = 1.f / sqrtf(fvar + eps);
    save_mean[c] = fmean;
    save_invstd[c] = invstd;
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    scale[c] = g * invstd;
    shift[c] = b - fmean * (g * invstd);
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * fmean;
    if (running_var) {
        const float unbiased = (float)(n > 1.0 ? m2 / (n - 1.0) : var);       // torch: running_var tracks the unbiased estimate
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ z, int zcs, long long M, int C, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, int relu, T* __restrict__ y, int ycs) {
    const int G = C / BN_V;
    const long long total = M * G;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)blockDim.x * gridDim.x) {
        const long long m = idx / G;
        const int c = (int)(idx % G) * BN_V;
        float v[BN_V];
        bn_load4<T>(z + (size_t)m * zcs + c, v);
        const f32x4 sc = *(const f32x4*)(scale + c), sh = *(const f32x4*)(shift + c);
#pragma unroll
        for (int e = 0; e < BN_V; ++e) {
            v[e] = v[e] * sc[e] + sh[e];
            if (relu) v[e] = fmaxf(v[e], 0.f);
        }
        bn_store4<T>(y + (size_t)m * ycs + c, v);
    }
}

// backward pass 1: per chunk and channel  sum g  and  sum g * (z - mean),  g = gy * (y > 0)
template <typename T, typename TG>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ z, int zcs, const T* __restrict__ y, int ycs,
                                                            const TG* __restrict__ gy, int gcs, long long M, int C, int relu,
                                                            const float* __restrict__ save_mean, float* __restrict__ part) {
    __shared__ float red[2][256][BN_V];
    const long long p0 = (long long)blockIdx.x * BN_CHUNK;
    const int n = (int)((M - p0) < BN_CHUNK ? (M - p0) : BN_CHUNK);
    const BnLayout L = bn_layout(C);
    const int tid = threadIdx.x, cg = tid % L.cgs, pr = tid / L.cgs;
    for (int c0 = 0; c0 < C; c0 += L.cgs * BN_V) {
        const int c = c0 + cg * BN_V;
        const bool live = pr < L.rows && c < C;
        float s[BN_V] = {0.f, 0.f, 0.f, 0.f}, sx[BN_V] = {0.f, 0.f, 0.f, 0.f};
        if (live) {
            const f32x4 mu = *(const f32x4*)(save_mean + c);
            for (int i = pr; i < n; i += L.rows) {
                float zv[BN_V], yv[BN_V], gv[BN_V];
                bn_load4<T>(z + (size_t)(p0 + i) * zcs + c, zv);
                bn_load4<TG>(gy + (size_t)(p0 + i) * gcs + c, gv);
                if (relu) bn_load4<T>(y + (size_t)(p0 + i) * ycs + c, yv);
#pragma unroll
                for (int e = 0; e < BN_V; ++e) {
                    const float g = (relu && !(yv[e] > 0.f)) ? 0.f : gv[e];
                    s[e] += g;
                    sx[e] += g * (zv[e] - mu[e]);
                }
            }
        }
#pragma unroll
        for (int e = 0; e < BN_V; ++e) { red[0][tid][e] = s[e]; red[1][tid][e] = sx[e]; }
        __syncthreads();
        if (pr == 0 && c < C) {
#pragma unroll
            for (int e = 0; e < BN_V; ++e) {
                float S = 0.f, SX = 0.f;
                for (int r = 0; r < L.rows; ++r) { S += red[0][r * L.cgs + cg][e]; SX += red[1][r * L.cgs + cg][e]; }
                part[((size_t)blockIdx.x * C + c + e) * 2 + 0] = S;
                part[((size_t)blockIdx.x * C + c + e) * 2 + 1] = SX;
            }
        }
        __syncthreads();
    }
}

// one thread per channel: the chunk sums in chunk order (double) -> gbeta, ggamma and the two coefficients of the apply pass
//   gz = a * g - b - (z - mean) * d      a = gamma * invstd,  b = a * gbeta / M,  d = a * invstd^2 * sum(g (z - mean)) / M
__global__ void bn_bwd_finish_kernel(const float* __restrict__ part, int chunks, long long M, int C, const float* __restrict__ gamma,
                                     const float* __restrict__ save_invstd, float* __restrict__ ggamma, float* __restrict__ gbeta,
                                     float* __restrict__ coef) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0.0, sx = 0.0;
    for (int k = 0; k < chunks; ++k) { s += (double)part[((size_t)k * C + c) * 2]; sx += (double)part[((size_t)k * C + c) * 2 + 1]; }
    const float invstd = save_invstd[c], g = gamma ? gamma[c] : 1.f;
    if (gbeta) gbeta[c] = (float)s;
    if (ggamma) ggamma[c] = (float)(sx * (double)invstd);
    const float a = g * invstd;
    coef[3 * c + 0] = a;
    coef[3 * c + 1] = (float)((double)a * s / (double)M);
    coef[3 * c + 2] = (float)((double)a * (double)invstd * (double)invstd * sx / (double)M);
}

template <typename T, typename TG>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ z, int zcs, const T* __restrict__ y, int ycs,
                                                           const TG* __restrict__ gy, int gcs, long long M, int C, int relu,
                                                           const float* __restrict__ save_mean, const float* __restrict__ coef,
                                                           T* __restrict__ gz) {
    const int G = C / BN_V;
    const long long total = M * G;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)blockDim.x * gridDim.x) {
        const long long m = idx / G;
        const int c = (int)(idx % G) * BN_V;
        float zv[BN_V], yv[BN_V], gv[BN_V], o[BN_V];
        bn_load4<T>(z + (size_t)m * zcs + c, zv);
        bn_load4<TG>(gy + (size_t)m * gcs + c, gv);
        if (relu) bn_load4<T>(y + (size_t)m * ycs + c, yv);
#pragma unroll
        for (int e = 0; e < BN_V; ++e) {
            const float g = (relu && !(yv[e] > 0.f)) ? 0.f : gv[e];
            o[e] = coef[3 * (c + e)] * g - coef[3 * (c + e) + 1] - (zv[e] - save_mean[c + e]) * coef[3 * (c + e) + 2];
        }
        bn_store4<T>(gz + (size_t)m * C + c, o);
    }
}

static inline int bn_chunks(long long M) { return (int)((M + BN_CHUNK - 1) / BN_CHUNK); }
static inline unsigned bn_flat_grid(long long total) {
    long long g = (total + 255) / 256;
    if (g > 256 * 32) g = 256 * 32;
    return (unsigned)(g < 1 ? 1 : g);
}

template <typename T>
static int bn_forward_t(const void* z, int zcs, long long M, int C, const float* gamma, const float* beta, float eps, float momentum,
                        float* rm, float* rv, float* save_mean, float* save_invstd, int relu, void* y, int ycs, float* ws,
                        step_stream_t stream) {
    const int chunks = bn_chunks(M);
    float* part = ws;
    float* scale = ws + (size_t)chunks * C * 2;
    float* shift = scale + C;
    STEP_LAUNCH((bn_stats_kernel<T>), dim3((unsigned)chunks), dim3(256), stream, (const T*)z, zcs, M, C, part);
    STEP_LAUNCH(bn_finish_kernel, dim3((unsigned)((C + 63) / 64)), dim3(64), stream, (const float*)part, chunks, M, C, gamma, beta, eps, momentum,
                rm, rv, save_mean, save_invstd, scale, shift);
    STEP_LAUNCH((bn_apply_kernel<T>), dim3(bn_flat_grid(M * (C / BN_V))), dim3(256), stream, (const T*)z, zcs, M, C, (const float*)scale,
                (const float*)shift, relu, (T*)y, ycs);
    return STEP_LAUNCH_CHECK();
}

template <typename T, typename TG>
static int bn_backward_t(const void* z, int zcs, const void* y, int ycs, const void* gy, int gcs, long long M, int C, int relu,
                         const float* gamma, const float* save_mean, const float* save_invstd, void* gz, float* ggamma, float* gbeta,
                         float* ws, step_stream_t stream) {
    const int chunks = bn_chunks(M);
    float* part = ws;
    float* coef = ws + (size_t)chunks * C * 2;
    STEP_LAUNCH((bn_bwd_reduce_kernel<T, TG>), dim3((unsigned)chunks), dim3(256), stream, (const T*)z, zcs, (const T*)y, ycs, (const TG*)gy, gcs, M,
                C, relu, save_mean, part);
    STEP_LAUNCH(bn_bwd_finish_kernel, dim3((unsigned)((C + 63) / 64)), dim3(64), stream, (const float*)part, chunks, M, C, gamma, save_invstd, ggamma,
                gbeta, coef);
    STEP_LAUNCH((bn_bwd_apply_kernel<T, TG>), dim3(bn_flat_grid(M * (C / BN_V))), dim3(256), stream, (const T*)z, zcs, (const T*)y, ycs,
                (const TG*)gy, gcs, M, C, relu, save_mean, (const float*)coef, (T*)gz);
    return STEP_LAUNCH_CHECK();
}

}  // namespace step

using namespace step;

extern "C" {

size_t step_bn_train_workspace_bytes(long long M, int C) {
    if (M <= 0 || C <= 0) return 0;
    return ((size_t)bn_chunks(M) * C * 2 + (size_t)3 * C + 16) * sizeof(float);
}

static int bn_check(int dtype, long long M, int C, int zcs, const void* z) {
    if (M < 0 || C <= 0 || zcs < C) return STEP_E_SHAPE;
    if ((C % BN_V) || (zcs % BN_V)) return STEP_E_UNSUPPORTED;
    if (dtype != STEP_F32 && dtype != STEP_BF16 && dtype != STEP_F16) return STEP_E_DTYPE;
    if ((uintptr_t)z % (dtype == STEP_F32 ? 16 : 8)) return STEP_E_ALIGN;
    return STEP_OK;
}

int step_bn_train_forward(int dtype, const void* z, int z_cstride, long long M, int C, const float* gamma, const float* beta, float eps,
                          float momentum, float* running_mean, float* running_var, float* save_mean, float* save_invstd, int relu, void* y,
                          int y_cstride, void* ws, size_t ws_bytes, step_stream_t stream) {
    if (z_cstride == 0) z_cstride = C;
    if (y_cstride == 0) y_cstride = C;
    int rc = bn_check(dtype, M, C, z_cstride, z);
    if (rc) return rc;
    if (y_cstride < C || (y_cstride % BN_V) || !(eps >= 0.f) || !(momentum >= 0.f && momentum <= 1.f)) return STEP_E_SHAPE;
    if (M == 0) return STEP_OK;
    if (!z || !y || !save_mean || !save_invstd || !ws) return STEP_E_NULL;
    if (ws_bytes < step_bn_train_workspace_bytes(M, C)) return STEP_E_SHAPE;
    if (((uintptr_t)ws % 16) || ((uintptr_t)y % (dtype == STEP_F32 ? 16 : 8))) return STEP_E_ALIGN;
    switch (dtype) {
        case STEP_F32: return bn_forward_t<float>(z, z_cstride, M, C, gamma, beta, eps, momentum, running_mean, running_var, save_mean, save_invstd, relu, y, y_cstride, (float*)ws, stream);
        case STEP_BF16: return bn_forward_t<bf16_t>(z, z_cstride, M, C, gamma, beta, eps, momentum, running_mean, running_var, save_mean, save_invstd, relu, y, y_cstride, (float*)ws, stream);
        case STEP_F16: return bn_forward_t<f16_t>(z, z_cstride, M, C, gamma, beta, eps, momentum, running_mean, running_var, save_mean, save_invstd, relu, y, y_cstride, (float*)ws, stream);
    }
    return STEP_E_DTYPE;
}

int step_bn_train_backward(int dtype, const void* z, int z_cstride, const void* y, int y_cstride, int gy_dtype, const void* gy, int gy_cstride,
                           long long M, int C, int relu, const float* gamma, const float* save_mean, const float* save_invstd, void* gz,
                           float* ggamma, float* gbeta, void* ws, size_t ws_bytes, step_stream_t stream) {
    if (z_cstride == 0) z_cstride = C;
    if (y_cstride == 0) y_cstride = C;
    if (gy_cstride == 0) gy_cstride = C;
    int rc = bn_check(dtype, M, C, z_cstride, z);
    if (rc) return rc;
    if (y_cstride < C || gy_cstride < C || (y_cstride % BN_V) || (gy_cstride % BN_V)) return STEP_E_SHAPE;
    if (gy_dtype != STEP_F32 && gy_dtype != dtype) return STEP_E_DTYPE;
    if (M == 0) return STEP_OK;
    if (!z || !gy || !gz || (relu && !y) || !save_mean || !save_invstd || !ws) return STEP_E_NULL;
    if (ws_bytes < step_bn_train_workspace_bytes(M, C)) return STEP_E_SHAPE;
    if (((uintptr_t)ws % 16) || ((uintptr_t)gy % (gy_dtype == STEP_F32 ? 16 : 8)) || ((uintptr_t)gz % (dtype == STEP_F32 ? 16 : 8)) ||
        (relu && ((uintptr_t)y % (dtype == STEP_F32 ? 16 : 8)))) return STEP_E_ALIGN;
    const bool gf = gy_dtype == STEP_F32;
#define BN_BWD(T, TG) bn_backward_t<T, TG>(z, z_cstride, y, y_cstride, gy, gy_cstride, M, C, relu, gamma, save_mean, save_invstd, gz, ggamma, gbeta, (float*)ws, stream)
    switch (dtype) {
        case STEP_F32: return BN_BWD(float, float);
        case STEP_BF16: return gf ? BN_BWD(bf16_t, float) : BN_BWD(bf16_t, bf16_t);
        case STEP_F16: return gf ? BN_BWD(f16_t, float) : BN_BWD(f16_t, f16_t);
    }
#undef BN_BWD
    return STEP_E_DTYPE;
}

}  // extern "C"
