// step_amd/csrc/pool.hip -- TF-"SAME" max pool, window average pool and layout transposes on
// channels-last activations for gfx950.
//
// Replaces MaxPool3dTFPadding (models/i3dpt.py:114-126: ConstantPad3d(0) copy + MaxPool3d with
// ceil_mode -- two full passes over the activation) with ONE pass: the TF pad is a predicate (a
// padded tap contributes the VALUE 0, exactly like the reference's explicit zero pad; a
// ceil-mode overhang beyond the pad is skipped), and nn.AvgPool3d((1,13,13)) of ContextNet
// (models/two_branch.py:127).  Both are pure HBM-bound streaming ops: lanes run along C in
// 16-byte vectors, so every tap is a fully coalesced row segment.
#include "common.h"

namespace step {

__host__ __device__ static inline int tf_pad_front(int k, int s) { int a = k - s; if (a < 0) a = 0; return a / 2; }
__host__ __device__ static inline int tf_pad_total(int k, int s) { int a = k - s; return a < 0 ? 0 : a; }
// models/i3dpt.py:114-126 + torch MaxPool3d(ceil_mode=True) on the explicitly padded input
__host__ __device__ static inline int pool_out_size(int L, int k, int s) {
    int Lp = L + tf_pad_total(k, s);
    int o = (Lp - k + s - 1) / s + 1;
    if ((o - 1) * s >= Lp) --o;
    return o;
}

template <typename T, int V> struct Vec16;
template <> struct Vec16<float, 4> {
    typedef f32x4 raw;
    __device__ static __forceinline__ void unpack(const raw& r, float (&f)[4]) { for (int i = 0; i < 4; ++i) f[i] = r[i]; }
    __device__ static __forceinline__ raw pack(const float (&f)[4]) { raw r = {f[0], f[1], f[2], f[3]}; return r; }
};
template <typename T> struct Vec16h {
    typedef u16x8 raw;
    __device__ static __forceinline__ void unpack(const raw& r, float (&f)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { T e; e.v = r[i]; f[i] = elem<T>::to_f32(e); }
    }
    __device__ static __forceinline__ raw pack(const float (&f)[8]) {
        raw r;
#pragma unroll
        for (int i = 0; i < 8; ++i) r[i] = elem<T>::from_f32(f[i]).v;
        return r;
    }
};
template <> struct Vec16<bf16_t, 8> : Vec16h<bf16_t> {};
template <> struct Vec16<f16_t, 8> : Vec16h<f16_t> {};

struct PoolParams {
    int N, D, H, W, C, x_cstride, x_coff;
    int Do, Ho, Wo, y_cstride, y_coff;
    int kd, kh, kw, sd, sh, sw;
    int pfd, pfh, pfw;        // TF front pads
    int Lpd, Lph, Lpw;        // padded extents
};

template <typename T>
__global__ void maxpool3d_tf_kernel(const T* __restrict__ x, T* __restrict__ y, PoolParams p, long long total) {
    constexpr int V = elem<T>::VEC;
    typedef typename Vec16<T, V>::raw raw;
    const int CV = p.C / V;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)blockDim.x * gridDim.x) {
        const int cv = (int)(idx % CV);
        long long pix = idx / CV;
        const int ow = (int)(pix % p.Wo); pix /= p.Wo;
        const int oh = (int)(pix % p.Ho); pix /= p.Ho;
        const int od = (int)(pix % p.Do);
        const int n = (int)(pix / p.Do);
        float m[V];
#pragma unroll
        for (int i = 0; i < V; ++i) m[i] = -FLT_MAX;
        for (int a = 0; a < p.kd; ++a) {
            const int pd = od * p.sd + a;
            if (pd >= p.Lpd) break;                 // ceil-mode overhang: ignored
            const int id = pd - p.pfd;
            for (int b = 0; b < p.kh; ++b) {
                const int phh = oh * p.sh + b;
                if (phh >= p.Lph) break;
                const int ih = phh - p.pfh;
                for (int c = 0; c < p.kw; ++c) {
                    const int pw = ow * p.sw + c;
                    if (pw >= p.Lpw) break;
                    const int iw = pw - p.pfw;
                    const bool inb = id >= 0 && id < p.D && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
                    if (inb) {
                        const T* src = x + ((((size_t)n * p.D + id) * p.H + ih) * p.W + iw) * p.x_cstride + p.x_coff + cv * V;
                        float f[V];
                        Vec16<T, V>::unpack(*(const raw*)src, f);
#pragma unroll
                        for (int i = 0; i < V; ++i) m[i] = fmaxf(m[i], f[i]);
                    } else {                          // explicit TF zero padding: value 0
#pragma unroll
                        for (int i = 0; i < V; ++i) m[i] = fmaxf(m[i], 0.f);
                    }
                }
            }
        }
        T* dst = y + ((((size_t)n * p.Do + od) * p.Ho + oh) * p.Wo + ow) * p.y_cstride + p.y_coff + cv * V;
        *(raw*)dst = Vec16<T, V>::pack(m);
    }
}

// 3x3x3, stride 1, TF-SAME (pad 1 each side, zero-VALUED) max pool -- the `branch_3` pool of every
// Inception block (models/i3dpt.py:151-155).
// A 256-thread workgroup owns an 8x16-pixel x 64-byte (32 x 16-bit / 16 x fp32 channels) column of the
// volume and walks along D with a rolling window of input planes in LDS (4 slots of [10][18] pixels: the
// three planes of the current window + the one being fetched).  Every input plane is read from global
// memory ONCE per column (1.4x the tile with its halo) instead of 27 times, the next plane's loads fly
// while the current outputs are computed from LDS, and every lane moves 16 bytes.
template <typename T>
__global__ __launch_bounds__(256) void maxpool333_s1_kernel(const T* __restrict__ x, T* __restrict__ y, PoolParams p,
                                                            int tiles_h, int tiles_w, int cchunks, int dseg) {
    constexpr int V = elem<T>::VEC;
    typedef typename Vec16<T, V>::raw raw;
    constexpr int TH = 8, TW = 16, HH = TH + 2, HW = TW + 2, SL = 4;       // SL 16-byte slots per pixel
    constexpr int PLANE = HH * HW * SL;                                      // vectors per plane (720)
    constexpr int NLD = (PLANE + 255) / 256;                                 // vectors per thread per plane (3)
    __shared__ __attribute__((aligned(16))) raw lds[4 * PLANE];

    const int tid = threadIdx.x;
    int t = blockIdx.x;
    const int cc = t % cchunks; t /= cchunks;
    const int tw_i = t % tiles_w; t /= tiles_w;
    const int th_i = t % tiles_h;
    const int n = t / tiles_h;
    const int h0 = th_i * TH, w0 = tw_i * TW;
    const int c0 = cc * SL * V;                                              // first channel of this chunk

    raw zero;
#pragma unroll
    for (int i = 0; i < V; ++i) zero[i] = 0;

    auto load_plane = [&](int d, raw (&r)[NLD]) {                            // global -> registers (zero outside)
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            const int v = tid + q * 256;
            raw val = zero;
            if (v < PLANE && d >= 0 && d < p.D) {
                const int pix = v / SL, sl = v % SL;
                const int ih = h0 + pix / HW - 1, iw = w0 + pix % HW - 1;
                const int c = c0 + sl * V;
                if (ih >= 0 && ih < p.H && iw >= 0 && iw < p.W && c < p.C)
                    val = *(const raw*)(x + ((((size_t)n * p.D + d) * p.H + ih) * p.W + iw) * p.x_cstride + p.x_coff + c);
            }
            r[q] = val;
        }
    };
    auto store_plane = [&](int slot, const raw (&r)[NLD]) {
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            const int v = tid + q * 256;
            if (v < PLANE) lds[slot * PLANE + v] = r[q];
        }
    };

    // this workgroup walks the planes [dbeg, dend) of its column (grid.y segments along D: more workgroups
    // in flight on the small maps; a segment re-reads one halo plane on each side)
    const int dbeg = blockIdx.y * dseg, dend = min(dbeg + dseg, p.D);
    raw r[NLD];
    load_plane(dbeg - 1, r); store_plane((dbeg - 1) & 3, r);
    load_plane(dbeg, r);     store_plane(dbeg & 3, r);
    load_plane(dbeg + 1, r);                        // in flight
    for (int d = dbeg; d < dend; ++d) {
        store_plane((d + 1) & 3, r);                // plane d+1 (zero plane when d+1 == D)
        __syncthreads();
        if (d + 2 <= p.D) load_plane(d + 2, r);     // next window's new plane flies during the compute below
        // outputs of plane d: 128 pixels x 4 vectors = 512 items, 2 per thread
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int item = tid + q * 256;
            const int sl = item % SL, pix = item / SL;
            const int oh = pix / TW, ow = pix % TW;
            float m[V];
#pragma unroll
            for (int i = 0; i < V; ++i) m[i] = -FLT_MAX;
#pragma unroll
            for (int a = -1; a <= 1; ++a) {
                const raw* pl = lds + ((d + a) & 3) * PLANE;
#pragma unroll
                for (int b = 0; b < 3; ++b)
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        float f[V];
                        Vec16<T, V>::unpack(pl[((oh + b) * HW + ow + c) * SL + sl], f);
#pragma unroll
                        for (int i = 0; i < V; ++i) m[i] = fmaxf(m[i], f[i]);
                    }
            }
            const int gh = h0 + oh, gw = w0 + ow, c = c0 + sl * V;
            if (gh < p.H && gw < p.W && c < p.C)
                *(raw*)(y + ((((size_t)n * p.D + d) * p.H + gh) * p.W + gw) * p.y_cstride + p.y_coff + c) = Vec16<T, V>::pack(m);
        }
        // no trailing barrier: the slot refilled next iteration ((d+2)&3 = plane d-2) was last read in
        // iteration d-1, and every thread has passed this iteration's barrier since
    }
}

template <typename T>
__global__ void avgpool_hw_kernel(const T* __restrict__ x, T* __restrict__ y, int ND, int H, int W, int C, int kh,
                                  int kw, long long total) {
    constexpr int V = elem<T>::VEC;
    typedef typename Vec16<T, V>::raw raw;
    const int CV = C / V;
    const int Ho = H - kh + 1, Wo = W - kw + 1;
    const float inv_div = (float)(kh * kw);
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)blockDim.x * gridDim.x) {
        const int cv = (int)(idx % CV);
        long long pix = idx / CV;
        const int ow = (int)(pix % Wo); pix /= Wo;
        const int oh = (int)(pix % Ho);
        const int nd = (int)(pix / Ho);
        float s[V];
#pragma unroll
        for (int i = 0; i < V; ++i) s[i] = 0.f;
        for (int b = 0; b < kh; ++b)
            for (int c = 0; c < kw; ++c) {
                const T* src = x + (((size_t)nd * H + oh + b) * W + ow + c) * C + cv * V;
                float f[V];
                Vec16<T, V>::unpack(*(const raw*)src, f);
#pragma unroll
                for (int i = 0; i < V; ++i) s[i] += f[i];
            }
#pragma unroll
        for (int i = 0; i < V; ++i) s[i] = s[i] / inv_div;
        *(raw*)(y + (((size_t)nd * Ho + oh) * Wo + ow) * C + cv * V) = Vec16<T, V>::pack(s);
    }
}

// [N,C,S] <-> [N,S,C] through a 32x33 LDS tile (both sides coalesced).  256 threads, 32x32 tile.
template <typename TS, typename TD>
__global__ void transpose_cs_kernel(const TS* __restrict__ src, TD* __restrict__ dst, int C, long long S,
                                    int to_channels_last) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z;
    // rows = slow dim of the source, cols = fast dim of the source
    const long long R = to_channels_last ? C : S;       // source rows
    const long long Q = to_channels_last ? S : C;       // source cols (contiguous)
    const long long r0 = (long long)blockIdx.y * 32, q0 = (long long)blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const TS* s = src + (size_t)n * R * Q;
    TD* d = dst + (size_t)n * R * Q;
    for (int j = ty; j < 32; j += 8) {
        long long r = r0 + j, q = q0 + tx;
        tile[j][tx] = (r < R && q < Q) ? elem<TS>::to_f32(s[r * Q + q]) : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        long long q = q0 + j, r = r0 + tx;              // dst is [Q][R]
        if (r < R && q < Q) d[q * R + r] = elem<TD>::from_f32(tile[tx][j]);
    }
}

static inline unsigned flat_grid(long long total, int block) {
    long long g = ceil_div64(total, block);
    if (g > 16384) g = 16384;
    return (unsigned)g;
}

template <typename T>
static int maxpool_t(const void* x, void* y, const PoolParams& p, step_stream_t stream) {
    constexpr int V = elem<T>::VEC;
    if (p.C % V || p.x_cstride % V || p.x_coff % V || p.y_cstride % V || p.y_coff % V) return STEP_E_ALIGN;
    if (p.kd == 3 && p.kh == 3 && p.kw == 3 && p.sd == 1 && p.sh == 1 && p.sw == 1) {
        const int tiles_h = ceil_div(p.H, 8), tiles_w = ceil_div(p.W, 16), cchunks = ceil_div(p.C, 4 * V);
        const long long blocks = (long long)p.N * tiles_h * tiles_w * cchunks;
        if (blocks == 0) return STEP_OK;
        // split D until there are a few thousand workgroups (but keep >= 4 planes per segment)
        int nseg = 1;
        while (blocks * nseg < 2048 && p.D / (nseg * 2) >= 4) nseg *= 2;
        const int dseg = ceil_div(p.D, nseg);
        STEP_LAUNCH((maxpool333_s1_kernel<T>), dim3((unsigned)blocks, (unsigned)ceil_div(p.D, dseg)), dim3(256), stream, (const T*)x, (T*)y, p, tiles_h, tiles_w, cchunks, dseg);
        return STEP_LAUNCH_CHECK();
    }
    long long total = (long long)p.N * p.Do * p.Ho * p.Wo * (p.C / V);
    if (total == 0) return STEP_OK;
    STEP_LAUNCH((maxpool3d_tf_kernel<T>), dim3(flat_grid(total, 256)), dim3(256), stream, (const T*)x, (T*)y, p, total);
    return STEP_LAUNCH_CHECK();
}

template <typename T>
static int avgpool_t(const void* x, void* y, int N, int D, int H, int W, int C, int kh, int kw, step_stream_t stream) {
    constexpr int V = elem<T>::VEC;
    if (C % V) return STEP_E_ALIGN;
    long long total = (long long)N * D * (H - kh + 1) * (W - kw + 1) * (C / V);
    if (total == 0) return STEP_OK;
    STEP_LAUNCH((avgpool_hw_kernel<T>), dim3(flat_grid(total, 256)), dim3(256), stream, (const T*)x, (T*)y, N * D, H, W,
                C, kh, kw, total);
    return STEP_LAUNCH_CHECK();
}

template <typename TS, typename TD>
static int transpose_t(const void* src, void* dst, int N, int C, long long S, int tcl, step_stream_t stream) {
    const long long R = tcl ? C : S, Q = tcl ? S : C;
    dim3 grid((unsigned)ceil_div64(Q, 32), (unsigned)ceil_div64(R, 32), (unsigned)N);
    STEP_LAUNCH((transpose_cs_kernel<TS, TD>), grid, dim3(256), stream, (const TS*)src, (TD*)dst, C, S, tcl);
    return STEP_LAUNCH_CHECK();
}

}  // namespace step

using namespace step;

extern "C" {

int step_pool_out_size(int L, int k, int s) { return pool_out_size(L, k, s); }

int step_maxpool3d_tf(int dtype, const void* x, int N, int D, int H, int W, int C, int x_cstride, int x_coff, int kd,
                      int kh, int kw, int sd, int sh, int sw, void* y, int y_cstride, int y_coff,
                      step_stream_t stream) {
    if (N < 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0 || kd <= 0 || kh <= 0 || kw <= 0 || sd <= 0 || sh <= 0 || sw <= 0)
        return STEP_E_SHAPE;
    if (N == 0) return STEP_OK;
    if (!x || !y) return STEP_E_NULL;
    PoolParams p;
    p.N = N; p.D = D; p.H = H; p.W = W; p.C = C; p.x_cstride = x_cstride; p.x_coff = x_coff;
    p.Do = pool_out_size(D, kd, sd); p.Ho = pool_out_size(H, kh, sh); p.Wo = pool_out_size(W, kw, sw);
    p.y_cstride = y_cstride; p.y_coff = y_coff;
    p.kd = kd; p.kh = kh; p.kw = kw; p.sd = sd; p.sh = sh; p.sw = sw;
    p.pfd = tf_pad_front(kd, sd); p.pfh = tf_pad_front(kh, sh); p.pfw = tf_pad_front(kw, sw);
    p.Lpd = D + tf_pad_total(kd, sd); p.Lph = H + tf_pad_total(kh, sh); p.Lpw = W + tf_pad_total(kw, sw);
    switch (dtype) {
        case STEP_F32: return maxpool_t<float>(x, y, p, stream);
        case STEP_BF16: return maxpool_t<bf16_t>(x, y, p, stream);
        case STEP_F16: return maxpool_t<f16_t>(x, y, p, stream);
    }
    return STEP_E_DTYPE;
}

int step_avgpool_hw(int dtype, const void* x, int N, int D, int H, int W, int C, int kh, int kw, void* y,
                    step_stream_t stream) {
    if (N < 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0 || kh <= 0 || kw <= 0 || kh > H || kw > W) return STEP_E_SHAPE;
    if (N == 0) return STEP_OK;
    if (!x || !y) return STEP_E_NULL;
    switch (dtype) {
        case STEP_F32: return avgpool_t<float>(x, y, N, D, H, W, C, kh, kw, stream);
        case STEP_BF16: return avgpool_t<bf16_t>(x, y, N, D, H, W, C, kh, kw, stream);
        case STEP_F16: return avgpool_t<f16_t>(x, y, N, D, H, W, C, kh, kw, stream);
    }
    return STEP_E_DTYPE;
}

int step_transpose_cs(const void* src, int sdt, void* dst, int ddt, int N, int C, long long S, int tcl,
                      step_stream_t stream) {
    if (N < 0 || C <= 0 || S <= 0) return STEP_E_SHAPE;
    if (N == 0) return STEP_OK;
    if (!src || !dst) return STEP_E_NULL;
#define TR(A, B) return transpose_t<A, B>(src, dst, N, C, S, tcl, stream)
    if (sdt == STEP_F32 && ddt == STEP_F32) TR(float, float);
    if (sdt == STEP_F32 && ddt == STEP_BF16) TR(float, bf16_t);
    if (sdt == STEP_F32 && ddt == STEP_F16) TR(float, f16_t);
    if (sdt == STEP_BF16 && ddt == STEP_F32) TR(bf16_t, float);
    if (sdt == STEP_F16 && ddt == STEP_F32) TR(f16_t, float);
    if (sdt == STEP_BF16 && ddt == STEP_BF16) TR(bf16_t, bf16_t);
    if (sdt == STEP_F16 && ddt == STEP_F16) TR(f16_t, f16_t);
#undef TR
    return STEP_E_DTYPE;
}

}  // extern "C"
