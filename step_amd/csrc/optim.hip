// step_amd/csrc/optim.hip -- fused multi-tensor Adam over a flat fp32 parameter arena.
//
// Replaces optimizer.step() of torch.optim.Adam(params, lr=args.det_lr) (reference train.py:126,348) over the 159+
// single-tensor parameter groups utils/solver.py:12-93 builds (per-group lr and weight_decay): the reference runs
// a handful of element-wise kernels per group; here all parameters, gradients and both moments live in four flat
// arenas and ONE launch updates them.  Pure HBM streaming: 16 B read + 12 B written per element (+4 B when the
// gradient is cleared in the same pass), no reuse -> bound by HBM bandwidth; 16-byte vectors, grid-stride.
//
// Arithmetic follows torch/optim/adam.py::_single_tensor_adam (amsgrad=False, maximize=False):
//   g  = grad * grad_scale (+ weight_decay * p)
//   m += (g - m) * (1 - beta1)                 (Tensor.lerp_)
//   v  = v * beta2 + (1 - beta2) * g * g
//   p -= (lr / (1 - beta1^t)) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
#include "common.h"
#include <cmath>
#include <cstdlib>

namespace step {

constexpr int ADAM_MAX_SEG = 4096;      // 32 KiB of LDS for the segment table

template <int MAXSEG>
__global__ __launch_bounds__(256) void adam_flat_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, long long nvec,
                                                        const long long* __restrict__ seg_end, const float* __restrict__ seg_lr,
                                                        const float* __restrict__ seg_wd, int n_seg, float beta2, float omb1,
                                                        float omb2, float eps, float bc1, float bc2_sqrt, float gscale,
                                                        int zero_grad, const float* __restrict__ bc_dev, const float* __restrict__ amp) {
    __shared__ long long s_end[MAXSEG];                    // 4 KiB for the usual few hundred tensors: LDS does not limit occupancy
    if (bc_dev) { bc1 = bc_dev[0]; bc2_sqrt = bc_dev[1]; } // step counter on the device (adam_bias_kernel): graph replays advance it
    // dynamic loss scaling (step_adam_flat_amp): amp = {scale, growth_tracker, found_inf, -}.  The gradients carry the factor `scale`;
    // a step whose gradients held an inf / nan is SKIPPED (apex amp O1's patched optimizer.step, torch.amp.GradScaler.step) -- no
    // moment decay, no parameter change; the gradient arena is still cleared when the caller asked for it.
    bool skip = false;
    if (amp) { skip = amp[2] != 0.f; gscale = gscale * (1.f / amp[0]); }
    for (int i = threadIdx.x; i < n_seg; i += blockDim.x) s_end[i] = seg_end[i];
    __syncthreads();
    for (long long vec = (long long)blockIdx.x * blockDim.x + threadIdx.x; vec < nvec; vec += (long long)blockDim.x * gridDim.x) {
        const long long e = vec * 4;
        int lo = 0, hi = n_seg - 1;                       // first segment whose end lies beyond e
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (s_end[mid] > e) hi = mid; else lo = mid + 1;
        }
        if (skip) {
            if (zero_grad) *(f32x4*)(g + e) = f32x4{0.f, 0.f, 0.f, 0.f};
            continue;
        }
        const float lr = seg_lr[lo], wd = seg_wd[lo];
        const float step_size = lr / bc1;
        f32x4 P = *(const f32x4*)(p + e), G = *(const f32x4*)(g + e), M = *(const f32x4*)(m + e), V = *(const f32x4*)(v + e);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float gj = G[j] * gscale;
            if (wd != 0.f) gj = gj + wd * P[j];
            const float mj = M[j] + (gj - M[j]) * omb1;
            const float vj = V[j] * beta2 + omb2 * gj * gj;
            const float denom = sqrtf(vj) / bc2_sqrt + eps;
            P[j] = P[j] - step_size * (mj / denom);
            M[j] = mj;
            V[j] = vj;
        }
        *(f32x4*)(p + e) = P;
        *(f32x4*)(m + e) = M;
        *(f32x4*)(v + e) = V;
        if (zero_grad) *(f32x4*)(g + e) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

// the step counter of a CAPTURED optimizer step lives on the device: a replayed graph cannot receive a new host scalar.  One
// thread advances it and leaves the two bias corrections (double precision, as the host path) for adam_flat_kernel.
__global__ void adam_bias_kernel(long long* step, float* bc, double beta1, double beta2, const float* amp) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (amp && amp[2] != 0.f) return;                  // skipped step (overflowed gradients): the step count does not advance
        const long long t = *step + 1;
        *step = t;
        bc[0] = (float)(1.0 - pow(beta1, (double)t));
        bc[1] = (float)sqrt(1.0 - pow(beta2, (double)t));
    }
}

// ---- dynamic loss scaling (mixed-precision training: train.py:136-139 `amp.initialize(opt_level="O1")`, :342-345 `amp.scale_loss`) ----
// amp_state = {scale, growth_tracker, found_inf, unused}: the state of apex's DynamicLossScaler / torch.amp.GradScaler, on the device so
// that a captured training step needs no host decision.  grad_scan raises found_inf when any gradient is inf / nan (the overflow check
// apex runs while unscaling); loss_scale_update is GradScaler.update(): overflow -> scale *= backoff, tracker = 0; else tracker += 1 and
// at growth_interval clean steps scale *= growth, tracker = 0; found_inf is cleared for the next step.
__global__ __launch_bounds__(256) void grad_scan_kernel(const float* __restrict__ g, long long nvec, float* __restrict__ amp) {
    bool bad = false;
    for (long long vec = (long long)blockIdx.x * blockDim.x + threadIdx.x; vec < nvec; vec += (long long)blockDim.x * gridDim.x) {
        const u32x4 v = *(const u32x4*)(g + vec * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) bad |= (v[j] & 0x7f800000u) == 0x7f800000u;     // exponent all ones: inf or nan
    }
    if (bad) amp[2] = 1.f;                                  // (every writer stores the same value)
}

__global__ void loss_scale_update_kernel(float* amp, float growth, float backoff, int interval) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (amp[2] != 0.f) { amp[0] = amp[0] * backoff; amp[1] = 0.f; }
        else {
            const float t = amp[1] + 1.f;
            if (t >= (float)interval) { amp[0] = amp[0] * growth; amp[1] = 0.f; }
            else amp[1] = t;
        }
        amp[2] = 0.f;
    }
}

// ---- activation gradient of the fused conv unit ---------------------------------------------------------------------
// backward of  y = relu(conv * scale[c] + shift[c])  up to the conv:  g = gy * (y > 0) * scale[c], written once as fp32
// (operand of the weight-gradient kernel) and / or once in the activation dtype (operand of the data-gradient conv).
// One read of y and gy instead of the cast / compare / multiply / multiply / cast chain of element-wise passes.
typedef unsigned short u16x4_t __attribute__((ext_vector_type(4)));
template <typename T> struct Vec4;
template <> struct Vec4<float> {
    __device__ static __forceinline__ f32x4 load(const float* p) { return *(const f32x4*)p; }
    __device__ static __forceinline__ void store(float* p, const f32x4& v) { *(f32x4*)p = v; }
};
template <typename T> struct Vec4 {
    __device__ static __forceinline__ f32x4 load(const T* p) {
        const u16x4_t r = *(const u16x4_t*)p;
        return f32x4{elem<T>::from_bits16(r[0]), elem<T>::from_bits16(r[1]), elem<T>::from_bits16(r[2]), elem<T>::from_bits16(r[3])};
    }
    __device__ static __forceinline__ void store(T* p, const f32x4& v) {
        *(u16x4_t*)p = u16x4_t{elem<T>::bits16(v[0]), elem<T>::bits16(v[1]), elem<T>::bits16(v[2]), elem<T>::bits16(v[3])};
    }
};

template <typename TY, typename TG>
__global__ __launch_bounds__(256) void act_grad_kernel(const TY* __restrict__ y, const TG* __restrict__ gy, const float* __restrict__ scale,
                                                       long long nvec, int C, int y_cs, int gy_cs, int relu, float* __restrict__ g32,
                                                       TY* __restrict__ gt) {
    const int cv = C >> 2;                                                   // C % 4 == 0: a vector never straddles two pixels
    for (long long vec = (long long)blockIdx.x * blockDim.x + threadIdx.x; vec < nvec; vec += (long long)blockDim.x * gridDim.x) {
        const long long m = vec / cv;                                        // pixel; y / gy may be channel slices of wider buffers
        const int c = (int)(vec - m * cv) << 2;
        const long long e = vec * 4;                                         // the outputs are dense [M, C]
        f32x4 g = Vec4<TG>::load(gy + m * gy_cs + c);
        if (scale) {
            const f32x4 s = *(const f32x4*)(scale + c);
            g = f32x4{g[0] * s[0], g[1] * s[1], g[2] * s[2], g[3] * s[3]};
        }
        if (relu) {
            const f32x4 yv = Vec4<TY>::load(y + m * y_cs + c);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (!(yv[j] > 0.f)) g[j] = 0.f;
        }
        if (g32) *(f32x4*)(g32 + e) = g;
        if (gt) Vec4<TY>::store(gt + e, g);
    }
}

// 16-bit activations, C % 8 == 0: a 16-byte vector per lane and iteration, the (pixel, channel) pair of the grid-stride walk carried
// incrementally (the 4-wide form above pays a 64-bit division per 8 bytes: measured 1.6 TB/s of operand traffic over the 72 calls of
// a training step).  Same arithmetic per element -- fp32 product, mask, one rounding -- so the two forms agree bit for bit.
template <typename TY, typename TG>
__global__ __launch_bounds__(256) void act_grad8_kernel(const TY* __restrict__ y, const TG* __restrict__ gy, const float* __restrict__ scale,
                                                        long long nvec, int C, int y_cs, int gy_cs, int relu, float* __restrict__ g32,
                                                        TY* __restrict__ gt) {
    static_assert(sizeof(TY) == 2, "16-bit activations");
    const int cv = C >> 3;
    const long long stride = (long long)blockDim.x * gridDim.x;
    long long vec = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (vec >= nvec) return;
    long long m = vec / cv;
    int c = (int)(vec - m * cv);
    const long long dm = stride / cv;
    const int dc = (int)(stride - dm * cv);
    for (; vec < nvec; vec += stride) {
        float g[8];
        const TG* gp = gy + m * gy_cs + c * 8;
        if constexpr (sizeof(TG) == 4) {
            const f32x4 a = *(const f32x4*)gp, b = *(const f32x4*)(gp + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { g[j] = a[j]; g[j + 4] = b[j]; }
        } else {
            const u16x8 r = *(const u16x8*)gp;
#pragma unroll
            for (int j = 0; j < 8; ++j) g[j] = elem<TG>::from_bits16(r[j]);
        }
        if (scale) {
            const f32x4 s0 = *(const f32x4*)(scale + c * 8), s1 = *(const f32x4*)(scale + c * 8 + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { g[j] *= s0[j]; g[j + 4] *= s1[j]; }
        }
        if (relu) {
            const u16x8 yr = *(const u16x8*)(y + m * y_cs + c * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (!(elem<TY>::from_bits16(yr[j]) > 0.f)) g[j] = 0.f;
        }
        if (g32) {
            *(f32x4*)(g32 + vec * 8) = f32x4{g[0], g[1], g[2], g[3]};
            *(f32x4*)(g32 + vec * 8 + 4) = f32x4{g[4], g[5], g[6], g[7]};
        }
        if (gt) {
            u16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = elem<TY>::bits16(g[j]);
            *(u16x8*)(gt + vec * 8) = o;
        }
        m += dm; c += dc;
        if (c >= cv) { c -= cv; ++m; }
    }
}

template <typename TY, typename TG>
static int act_grad_t(const void* y, int y_cs, const void* gy, int gy_cs, const float* scale, long long M, int C, int relu, float* g32, void* gt,
                      step_stream_t stream) {
    if constexpr (sizeof(TY) == 2) {
        const uintptr_t al = (relu ? (uintptr_t)y : 0) | (uintptr_t)gy | (uintptr_t)gt | (uintptr_t)g32 | (uintptr_t)scale;
        if (!(C & 7) && !(y_cs & 7) && !(gy_cs & 7) && !(al & 15)) {
            const long long nvec8 = M * C / 8;
            long long blocks8 = (nvec8 + 255) / 256;
            if (blocks8 > 256LL * 32) blocks8 = 256LL * 32;
#ifdef STEP_EMUL
            if (blocks8 > 2) blocks8 = 2;                  // (host emulator: let small cases walk the grid-stride loop and its carries)
#endif
            STEP_LAUNCH((act_grad8_kernel<TY, TG>), dim3((unsigned)blocks8), dim3(256), stream, (const TY*)y, (const TG*)gy, scale, nvec8, C, y_cs, gy_cs, relu,
                        g32, (TY*)gt);
            return STEP_LAUNCH_CHECK();
        }
    }
    const long long nvec = M * C / 4;
    long long blocks = (nvec + 255) / 256;
    if (blocks > 256LL * 64) blocks = 256LL * 64;
    STEP_LAUNCH((act_grad_kernel<TY, TG>), dim3((unsigned)blocks), dim3(256), stream, (const TY*)y, (const TG*)gy, scale, nvec, C, y_cs, gy_cs, relu, g32,
                (TY*)gt);
    return STEP_LAUNCH_CHECK();
}

}  // namespace step

using namespace step;

extern "C" {

static int adam_flat_launch(float* param, float* grad, float* exp_avg, float* exp_avg_sq, long long n, const long long* seg_end,
                            const float* seg_lr, const float* seg_wd, int n_seg, double beta1, double beta2, double eps, int step_no,
                            long long* step_dev, float* bc_dev, float grad_scale, int zero_grad, step_stream_t stream,
                            const float* amp = nullptr) {
    if (n < 0 || (n & 3) || n_seg <= 0 || n_seg > ADAM_MAX_SEG || (!step_dev && step_no < 1)) return STEP_E_SHAPE;
    if (!(beta1 >= 0. && beta1 < 1.) || !(beta2 >= 0. && beta2 < 1.) || !(eps >= 0.)) return STEP_E_SHAPE;
    if (step_dev && !bc_dev) return STEP_E_NULL;
    if (step_dev) STEP_LAUNCH(adam_bias_kernel, dim3(1), dim3(64), stream, step_dev, bc_dev, beta1, beta2, amp);   // (also for an empty arena: the step counts)
    if (n == 0) return step_dev ? STEP_LAUNCH_CHECK() : STEP_OK;
    if (!param || !grad || !exp_avg || !exp_avg_sq || !seg_end || !seg_lr || !seg_wd) return STEP_E_NULL;
    if ((((uintptr_t)param) | ((uintptr_t)grad) | ((uintptr_t)exp_avg) | ((uintptr_t)exp_avg_sq)) & 15) return STEP_E_ALIGN;
    // the scalars are Python doubles in torch: 1 - beta is taken in double (1.f - 0.999f is off by 1.3e-5 relative)
    const float bc1 = step_dev ? 1.f : (float)(1.0 - std::pow(beta1, (double)step_no));
    const float bc2_sqrt = step_dev ? 1.f : (float)std::sqrt(1.0 - std::pow(beta2, (double)step_no));
    const float* bcd = step_dev ? bc_dev : nullptr;
    const long long nvec = n >> 2;
    long long blocks = (nvec + 255) / 256;
    constexpr int per_cu = 64;
    if (blocks > 256LL * per_cu) blocks = 256LL * per_cu; // 64 workgroups per CU (measured: 4.6 TB/s at 16, 5.5 TB/s at 64), grid-stride beyond
    if (n_seg <= 512)
        STEP_LAUNCH((adam_flat_kernel<512>), dim3((unsigned)blocks), dim3(256), stream, param, grad, exp_avg, exp_avg_sq, nvec, seg_end,
                    seg_lr, seg_wd, n_seg, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps, bc1, bc2_sqrt,
                    grad_scale, zero_grad, bcd, amp);
    else
        STEP_LAUNCH((adam_flat_kernel<ADAM_MAX_SEG>), dim3((unsigned)blocks), dim3(256), stream, param, grad, exp_avg, exp_avg_sq, nvec,
                    seg_end, seg_lr, seg_wd, n_seg, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps, bc1,
                    bc2_sqrt, grad_scale, zero_grad, bcd, amp);
    return STEP_LAUNCH_CHECK();
}

int step_adam_flat(float* param, float* grad, float* exp_avg, float* exp_avg_sq, long long n, const long long* seg_end,
                   const float* seg_lr, const float* seg_wd, int n_seg, double beta1, double beta2, double eps, int step_no,
                   float grad_scale, int zero_grad, step_stream_t stream) {
    return adam_flat_launch(param, grad, exp_avg, exp_avg_sq, n, seg_end, seg_lr, seg_wd, n_seg, beta1, beta2, eps, step_no, nullptr, nullptr,
                            grad_scale, zero_grad, stream);
}

int step_adam_flat_dev(float* param, float* grad, float* exp_avg, float* exp_avg_sq, long long n, const long long* seg_end,
                       const float* seg_lr, const float* seg_wd, int n_seg, double beta1, double beta2, double eps, long long* step_dev,
                       float* bias_corr, float grad_scale, int zero_grad, step_stream_t stream) {
    if (!step_dev || !bias_corr) return STEP_E_NULL;
    return adam_flat_launch(param, grad, exp_avg, exp_avg_sq, n, seg_end, seg_lr, seg_wd, n_seg, beta1, beta2, eps, 0, step_dev, bias_corr,
                            grad_scale, zero_grad, stream);
}

int step_adam_flat_amp(float* param, float* grad, float* exp_avg, float* exp_avg_sq, long long n, const long long* seg_end,
                       const float* seg_lr, const float* seg_wd, int n_seg, double beta1, double beta2, double eps, long long* step_dev,
                       float* bias_corr, float grad_scale, int zero_grad, float* amp_state, float growth_factor, float backoff_factor,
                       int growth_interval, step_stream_t stream) {
    if (!step_dev || !bias_corr || !amp_state) return STEP_E_NULL;
    if (n < 0 || (n & 3) || growth_interval < 1 || !(growth_factor >= 1.f) || !(backoff_factor > 0.f && backoff_factor <= 1.f)) return STEP_E_SHAPE;
    if (n > 0) {
        if (!grad) return STEP_E_NULL;
        if ((uintptr_t)grad & 15) return STEP_E_ALIGN;
        const long long nvec = n >> 2;
        long long blocks = (nvec + 255) / 256;
        if (blocks > 256LL * 32) blocks = 256LL * 32;
        STEP_LAUNCH(grad_scan_kernel, dim3((unsigned)blocks), dim3(256), stream, grad, nvec, amp_state);
    }
    const int rc = adam_flat_launch(param, grad, exp_avg, exp_avg_sq, n, seg_end, seg_lr, seg_wd, n_seg, beta1, beta2, eps, 0, step_dev, bias_corr,
                                    grad_scale, zero_grad, stream, amp_state);
    if (rc) return rc;
    STEP_LAUNCH(loss_scale_update_kernel, dim3(1), dim3(64), stream, amp_state, growth_factor, backoff_factor, growth_interval);
    return STEP_LAUNCH_CHECK();
}

int step_act_grad(int dtype, const void* y, int y_cstride, int gy_dtype, const void* gy, int gy_cstride, const float* scale, long long M, int C,
                  int relu, float* g32, void* g_act, step_stream_t stream) {
    if (M < 0 || C <= 0) return STEP_E_SHAPE;
    if (y_cstride == 0) y_cstride = C;
    if (gy_cstride == 0) gy_cstride = C;
    if (y_cstride < C || gy_cstride < C) return STEP_E_SHAPE;
    if ((C & 3) || (y_cstride & 3) || (gy_cstride & 3)) return STEP_E_UNSUPPORTED;     // 4-channel vectors
    {
        const int yb = dtype == STEP_F32 ? 16 : 8, gb = gy_dtype == STEP_F32 ? 16 : 8;
        if ((relu && ((uintptr_t)y & (yb - 1))) || ((uintptr_t)gy & (gb - 1)) || ((uintptr_t)g32 & 15) || ((uintptr_t)g_act & (yb - 1)) ||
            ((uintptr_t)scale & 15))
            return STEP_E_ALIGN;
    }
    if (M == 0) return STEP_OK;
    if (!gy || (relu && !y) || (!g32 && !g_act)) return STEP_E_NULL;
    if (gy_dtype != STEP_F32 && gy_dtype != dtype) return STEP_E_DTYPE;
    const bool gf = gy_dtype == STEP_F32;
    switch (dtype) {
        case STEP_F32: return act_grad_t<float, float>(y, y_cstride, gy, gy_cstride, scale, M, C, relu, g32, g_act, stream);
        case STEP_BF16: return gf ? act_grad_t<bf16_t, float>(y, y_cstride, gy, gy_cstride, scale, M, C, relu, g32, g_act, stream)
                                  : act_grad_t<bf16_t, bf16_t>(y, y_cstride, gy, gy_cstride, scale, M, C, relu, g32, g_act, stream);
        case STEP_F16: return gf ? act_grad_t<f16_t, float>(y, y_cstride, gy, gy_cstride, scale, M, C, relu, g32, g_act, stream)
                                 : act_grad_t<f16_t, f16_t>(y, y_cstride, gy, gy_cstride, scale, M, C, relu, g32, g_act, stream);
    }
    return STEP_E_DTYPE;
}

}  // extern "C"
