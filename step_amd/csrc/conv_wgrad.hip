// step_amd/csrc/conv_wgrad.hip -- weight gradients (training): conv_wgrad_kernel, stem_wgrad_kernel and their C entry
// points step_conv_wgrad / step_stem_wgrad.
#include "conv_common.h"

namespace step {

// ============================================================================================
// conv_wgrad_kernel -- weight gradient of a stride-1 SAME conv on channels-last tensors (train.py:257-348):
//     dW[co][ci][tap] = sum over pixels p of  dY[p][co] * X[p + tap][ci]
// A GEMM whose reduction axis is the PIXEL axis.  With channels innermost a lane's 16-bit MFMA fragment (8
// consecutive k for one row) would be a strided gather; the fp32 instruction v_mfma_f32_32x32x2_f32 takes ONE k
// per lane per issue, so with lanes along the channel axis every operand element is a plain coalesced load
// (32 consecutive channels of one pixel) -- no transposed copies, no LDS.  16-bit activations are widened on
// load; dY is fp32 (the epilogue's ReLU mask / BN scale are applied in fp32 by the caller).  Exact fp32 FMA
// chains per wavefront; partial sums of different wavefronts meet in fp32 atomics on dW.
//   wavefront job = one (n, d) plane (or one chunk of pixels of a pointwise layer) x one tap x one
//   (32*MB x 32*NB) tile of (co, ci); no barriers, four independent wavefronts per workgroup.
struct WgradParams {
    const void* x; const void* dy; float* dw;      // dy: fp32, or the activation type in the 16-bit-MFMA form (W16)
    int N, D, H, W, Cin, Cout, kd, kh, kw;
    int x_cstride, x_coff, dy_cstride, dy_coff;
    int cot, cit;                 // tiles along Cout / Cin
    int rows;                     // (n, d, h) rows per wavefront job
    long long total_rows;         // N * D * H
    long long jobs;               // ceil(total_rows / rows)
};

// W16 (16-bit storage only): dY arrives in the activation type too (what mixed-precision training back-propagates) and the
// products run on v_mfma_f32_32x32x16_{bf16,f16} -- 16x the fp32 instruction's rate -- with fp32 accumulation; the eight
// k values of a lane's fragment are still eight plain 2-byte loads of one channel at consecutive pixels (coalesced over the
// 32 lanes of a row block: 64-byte runs), packed pairwise by the loads themselves.  Same jobs, same atomics.
template <typename T, int MB, int NB, bool W16 = false>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradParams p) {
    static_assert(!W16 || sizeof(T) == 2, "the 16-bit-MFMA form needs 16-bit activations");
    typedef typename std::conditional<W16, T, float>::type DY;
    typedef typename std::conditional<W16, u16x8, f32x8>::type opfrag;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = lane & 31, khalf = lane >> 5;
    const long long job = (long long)blockIdx.x * 4 + wave;
    if (job >= p.jobs) return;                               // wave-uniform; the kernel has no barrier
    int t = blockIdx.y;
    const int cit_i = t % p.cit; t /= p.cit;
    const int cot_i = t % p.cot;
    const int tap = t / p.cot;
    const int ntaps = p.kd * p.kh * p.kw;
    const int kw_ = tap % p.kw, kh_ = (tap / p.kw) % p.kh, kd_ = tap / (p.kw * p.kh);
    const int co0 = cot_i * 32 * MB, ci0 = cit_i * 32 * NB;

    int coc[MB], cic[NB];
    bool cook[MB], ciok[NB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) { const int c = co0 + mb * 32 + m; cook[mb] = c < p.Cout; coc[mb] = cook[mb] ? c : p.Cout - 1; }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) { const int c = ci0 + nb * 32 + m; ciok[nb] = c < p.Cin; cic[nb] = ciok[nb] ? c : p.Cin - 1; }

    f32x16 acc[MB][NB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;

    const long long r_end = min((job + 1) * (long long)p.rows, p.total_rows);
    for (long long rr = job * (long long)p.rows; rr < r_end; ++rr) {
        const int h = (int)(rr % p.H);
        const long long plane = rr / p.H;
        const int n = (int)(plane / p.D), d = (int)(plane % p.D);
        const int id = d + kd_ - p.kd / 2, ih = h + kh_ - p.kh / 2;
        if (id < 0 || id >= p.D || ih < 0 || ih >= p.H) continue;        // this tap sees only zero padding from this row
        const DY* dyrow = (const DY*)p.dy + ((((size_t)n * p.D + d) * p.H + h) * p.W) * p.dy_cstride + p.dy_coff;
        const T* xrow = (const T*)p.x + ((((size_t)n * p.D + id) * p.H + ih) * p.W) * p.x_cstride + p.x_coff;
        for (int w0 = 0; w0 < p.W; w0 += 16) {
            opfrag a[MB], b[NB];
            auto ld_a = [&](size_t off) { if constexpr (W16) return dyrow[off].v; else return dyrow[off]; };
            auto ld_b = [&](size_t off) { if constexpr (W16) return xrow[off].v; else return elem<T>::to_f32(xrow[off]); };
            // Channel masks are not needed: rows / columns of a ragged (co, ci) tile read a clamped channel and their
            // accumulators are never stored.  Pixel masks matter only on the first / last step of a row.
            const int sh = kw_ - p.kw / 2;
            if (w0 + sh >= 0 && w0 + 16 + sh <= p.W && w0 + 16 <= p.W) {           // interior step (wave-uniform): plain loads
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int w = w0 + 8 * khalf + j;
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb) a[mb][j] = ld_a((size_t)w * p.dy_cstride + coc[mb]);
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) b[nb][j] = ld_b((size_t)(w + sh) * p.x_cstride + cic[nb]);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int w = w0 + 8 * khalf + j, iw = w + sh;
                    const bool aok = w < p.W, bok = aok && iw >= 0 && iw < p.W;
                    const int wc = aok ? w : p.W - 1, iwc = bok ? iw : 0;
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb) {
                        const auto v = ld_a((size_t)wc * p.dy_cstride + coc[mb]);
                        a[mb][j] = aok ? v : (decltype(v))0;
                    }
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        const auto v = ld_b((size_t)iwc * p.x_cstride + cic[nb]);
                        b[nb][j] = bok ? v : (decltype(v))0;
                    }
                }
            }
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) mma_k16(a[mb], b[nb], acc[mb][nb], typename std::conditional<W16, T, float>::type());
        }
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + mb * 32 + cd_row(r, lane), ci = ci0 + nb * 32 + (lane & 31);
                if (co < p.Cout && ci < p.Cin) atomicAdd(p.dw + ((size_t)co * p.Cin + ci) * ntaps + tap, acc[mb][nb][r]);
            }
}

// stem_wgrad_kernel -- weight gradient of the 7x7x7 stride-2 stem (Cin = 3) from the clip in its own [N,T,3,H,W]
// layout.  Same scheme as conv_wgrad_kernel (fp32 MFMA, reduction over output pixels, lanes along channels), but
// with only 3 input channels the B operand's 32 columns are the (kw, c) pairs of one (kd, kh) row of the filter
// (21 of 32 used): one wavefront job = one output plane (n, od) x one (kd, kh) x 64 output channels.
struct StemWgradParams {
    const void* x; const float* dy; float* dw;
    int N, T, H, W, To, Ho, Wo, Cout, cot;
    int rows, hchunks;            // output rows per job, jobs per output plane
    long long jobs;
};

template <typename T>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(StemWgradParams p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = lane & 31, khalf = lane >> 5;
    const long long job = (long long)blockIdx.x * 4 + wave;
    if (job >= p.jobs) return;
    int t = blockIdx.y;
    const int cot_i = t % p.cot; t /= p.cot;
    const int kh_ = t % 7, kd_ = t / 7;
    const int hc = (int)(job % p.hchunks);
    const long long plane = job / p.hchunks;
    const int n = (int)(plane / p.To), od = (int)(plane % p.To);
    const int it = 2 * od + kd_ - 2;
    if (it < 0 || it >= p.T) return;
    const int co0 = cot_i * 64;
    const int kw_ = m / 3, c_ = m % 3;
    const bool nok = m < 21;
    int coc[2]; bool cook[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) { const int c = co0 + mb * 32 + m; cook[mb] = c < p.Cout; coc[mb] = cook[mb] ? c : p.Cout - 1; }
    f32x16 acc[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;
    const T* xpl = (const T*)p.x + (((size_t)n * p.T + it) * 3 + (nok ? c_ : 0)) * p.H * p.W;
    for (int oh = hc * p.rows; oh < min((hc + 1) * p.rows, p.Ho); ++oh) {
        const int ih = 2 * oh + kh_ - 2;
        if (ih < 0 || ih >= p.H) continue;
        const float* dyrow = p.dy + ((((size_t)n * p.To + od) * p.Ho + oh) * p.Wo) * p.Cout;
        const T* xrow = xpl + (size_t)ih * p.W;
        for (int w0 = 0; w0 < p.Wo; w0 += 16) {
            f32x8 a[2], b;
            // interior step (wave-uniform): all 16 output pixels exist and all 7 taps of each fall inside the row ->
            // plain loads (channel / (kw, c) masks are unnecessary: surplus accumulator rows and columns are never stored)
            if (w0 + 16 <= p.Wo && 2 * w0 - 2 >= 0 && 2 * (w0 + 15) + 4 < p.W) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int ow = w0 + 8 * khalf + j;
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb) a[mb][j] = dyrow[(size_t)ow * p.Cout + coc[mb]];
                    b[j] = elem<T>::to_f32(xrow[2 * ow + (nok ? kw_ : 0) - 2]);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int ow = w0 + 8 * khalf + j, iw = 2 * ow + kw_ - 2;
                    const bool aok = ow < p.Wo, bok = aok && nok && iw >= 0 && iw < p.W;
                    const int owc = aok ? ow : p.Wo - 1, iwc = bok ? iw : 0;
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb) {
                        const float v = dyrow[(size_t)owc * p.Cout + coc[mb]];
                        a[mb][j] = aok ? v : 0.f;
                    }
                    const float xv = elem<T>::to_f32(xrow[iwc]);
                    b[j] = bok ? xv : 0.f;
                }
            }
            mma_k16(a[0], b, acc[0], float());
            mma_k16(a[1], b, acc[1], float());
        }
    }
    const int nn = lane & 31;
    if (nn < 21) {
        const int kw2 = nn / 3, c2 = nn % 3;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + mb * 32 + cd_row(r, lane);
                if (co < p.Cout) atomicAdd(p.dw + ((((size_t)co * 3 + c2) * 7 + kd_) * 7 + kh_) * 7 + kw2, acc[mb][r]);
            }
    }
}


}  // namespace step

using namespace step;

extern "C" {

// Every wavefront job ends in one set of fp32 atomics on its 64x64 (x tap) tile of dw: ~0.2 us per job at the rate the
// L2 sustains (measured: 5832 jobs on a 7x7 head layer = 1.16 ms for 2.9 GFLOP), against ~0.13 us of MFMA work per pixel
// of a job.  So a job must cover some hundred pixels or the launch is bound by the atomics, however small the map:
// the head layers on 7x7 maps ran at 0.3-4 TFLOP/s with the fixed ~6000-job split.  Swept on the C4 step (143 wgrad
// launches, ms in total): 16 px -> 37.4, 128 -> 24.0, 256 -> 21.5, 512 -> 22.1, 720 -> 22.7, 1440 -> 28.0, 5760 -> 43.5.
static int wgrad_min_pixels() {
    const char* e = getenv("STEP_WGRAD_MINPIX");               // read per call (tests exercise both regimes in one process)
    const int x = e ? atoi(e) : 0;
    return x > 0 ? (x + 15) / 16 * 16 : 512;
}

static int conv_wgrad_impl(const step_conv_desc* d, const void* x, const void* dy, bool w16, float* dw, int accumulate, step_stream_t stream) {
    if (!d) return STEP_E_NULL;
    if (w16 && d->dtype != STEP_BF16 && d->dtype != STEP_F16) return STEP_E_UNSUPPORTED;
    if (d->N < 0 || d->D <= 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->Cout <= 0) return STEP_E_SHAPE;
    if (d->kd <= 0 || d->kh <= 0 || d->kw <= 0 || !(d->kd & 1) || !(d->kh & 1) || !(d->kw & 1)) return STEP_E_UNSUPPORTED;
    if (d->x_coff < 0 || d->x_coff + d->Cin > d->x_cstride || d->y_coff < 0 || d->y_coff + d->Cout > d->y_cstride) return STEP_E_SHAPE;
    if (!dw) return STEP_E_NULL;
    const int ntaps = d->kd * d->kh * d->kw;
    if (!accumulate) {
        const int e = (int)hipMemsetAsync(dw, 0, (size_t)d->Cout * d->Cin * ntaps * sizeof(float), (hipStream_t)stream);
        if (e != 0) return e;
    }
    if (d->N == 0) return STEP_OK;
    if (!x || !dy) return STEP_E_NULL;
    WgradParams p;
    p.x = x; p.dy = dy; p.dw = dw;
    p.N = d->N; p.D = d->D; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Cout = d->Cout; p.kd = d->kd; p.kh = d->kh; p.kw = d->kw;
    p.x_cstride = d->x_cstride; p.x_coff = d->x_coff; p.dy_cstride = d->y_cstride; p.dy_coff = d->y_coff;
    if (ntaps == 1) {
        // pointwise: no neighbourhood, so the pixel axis is cut into chunks of 1024 ("rows" of one long plane list)
        const long long M = (long long)d->N * d->D * d->H * d->W;
        if (M > 0x7fffffffLL) return STEP_E_UNSUPPORTED;
        // pixels per wavefront job: ~6000 jobs per launch (see below), a multiple of the 16-pixel MFMA step
        static const int wg_jobs_pw = getenv("STEP_WGRAD_JOBS") ? atoi(getenv("STEP_WGRAD_JOBS")) : 6144;
        const long long tiles = (long long)ceil_div(d->Cout, 64) * ceil_div(d->Cin, d->Cin <= 32 ? 32 : 64);
        long long want = wg_jobs_pw / (tiles > 0 ? tiles : 1);
        if (want < 1) want = 1;
        long long ch = (ceil_div64(M, want) + 15) / 16 * 16;
        if (ch < wgrad_min_pixels()) ch = wgrad_min_pixels();
        if (ch > 65536) ch = 65536;
        const int chunk = (int)ch;
        // (n, d, h) collapse into full chunks; the ragged tail is a second launch
        const long long full = M / chunk;
        const int tail = (int)(M % chunk);
        int rc = STEP_OK;
        auto launch = [&](long long jobs, int W, size_t pix0) {
            p.N = 1; p.D = (int)jobs; p.H = 1; p.W = W; p.jobs = jobs; p.rows = 1; p.total_rows = jobs;
            p.x = (const char*)x + pix0 * d->x_cstride * (d->dtype == STEP_F32 ? 4 : 2);
            p.dy = (const char*)dy + pix0 * d->y_cstride * (w16 ? 2 : 4);
            const bool narrow = d->Cin <= 32;
            p.cot = ceil_div(d->Cout, 64); p.cit = ceil_div(d->Cin, narrow ? 32 : 64);
            dim3 grid((unsigned)ceil_div64(jobs, 4), (unsigned)(p.cot * p.cit));
#define STEP_WG(T_) do { if (narrow) STEP_LAUNCH((conv_wgrad_kernel<T_, 2, 1>), grid, dim3(256), stream, p); \
                         else STEP_LAUNCH((conv_wgrad_kernel<T_, 2, 2>), grid, dim3(256), stream, p); } while (0)
#define STEP_WG16(T_) do { if (narrow) STEP_LAUNCH((conv_wgrad_kernel<T_, 2, 1, true>), grid, dim3(256), stream, p); \
                           else STEP_LAUNCH((conv_wgrad_kernel<T_, 2, 2, true>), grid, dim3(256), stream, p); } while (0)
            switch (d->dtype) {
                case STEP_F32: STEP_WG(float); break;
                case STEP_BF16: if (w16) STEP_WG16(bf16_t); else STEP_WG(bf16_t); break;
                case STEP_F16: if (w16) STEP_WG16(f16_t); else STEP_WG(f16_t); break;
                default: rc = STEP_E_DTYPE;
            }
        };
        if (full) launch(full, chunk, 0);
        if (rc == STEP_OK && tail) launch(1, tail, (size_t)full * chunk);
        return rc != STEP_OK ? rc : STEP_LAUNCH_CHECK();
    }
    const bool narrow = d->Cin <= 32;
    p.cot = ceil_div(d->Cout, 64); p.cit = ceil_div(d->Cin, narrow ? 32 : 64);
    const long long gy = (long long)ntaps * p.cot * p.cit;
    // (n, d, h) rows per wavefront job.  Two opposite pressures (PMC): the kernel hides its load latency only with
    // several wavefronts per SIMD (1.6 per SIMD -> matrix pipe 18 % busy), but every job ends in one set of fp32
    // atomics (a 64x64 tile = 4096 of them; the 14x14 layers spent their time in 81 M atomics with one job per
    // plane).  Aim at ~6000 wavefront jobs per launch, whatever the map size.
    static const int wg_jobs = getenv("STEP_WGRAD_JOBS") ? atoi(getenv("STEP_WGRAD_JOBS")) : 6144;
    p.total_rows = (long long)d->N * d->D * d->H;
    {
        long long want = wg_jobs / (gy > 0 ? gy : 1);
        if (want < 1) want = 1;
        long long rows = ceil_div64(p.total_rows, want);
        const long long rows_min = ceil_div64(wgrad_min_pixels(), d->W);      // small maps: fewer, longer jobs (see above)
        if (rows < rows_min) rows = rows_min;
        if (rows > p.total_rows) rows = p.total_rows;
        if (rows < 1) rows = 1;
        if (rows > 0x3fffffff) rows = 0x3fffffff;
        p.rows = (int)rows;
    }
    p.jobs = ceil_div64(p.total_rows, p.rows);
    if (gy > 65535) return STEP_E_UNSUPPORTED;
    dim3 grid((unsigned)ceil_div64(p.jobs, 4), (unsigned)gy);
    switch (d->dtype) {
        case STEP_F32: STEP_WG(float); break;
        case STEP_BF16: if (w16) STEP_WG16(bf16_t); else STEP_WG(bf16_t); break;
        case STEP_F16: if (w16) STEP_WG16(f16_t); else STEP_WG(f16_t); break;
        default: return STEP_E_DTYPE;
    }
#undef STEP_WG
#undef STEP_WG16
    return STEP_LAUNCH_CHECK();
}

int step_conv_wgrad(const step_conv_desc* d, const void* x, const float* dy, float* dw, int accumulate, step_stream_t stream) {
    return conv_wgrad_impl(d, x, dy, false, dw, accumulate, stream);
}

int step_conv_wgrad16(const step_conv_desc* d, const void* x, const void* dy, float* dw, int accumulate, step_stream_t stream) {
    return conv_wgrad_impl(d, x, dy, true, dw, accumulate, stream);
}


int step_stem_wgrad(int dtype, const void* x, int N, int T, int H, int W, const float* dy, int Cout, float* dw, int accumulate,
                    step_stream_t stream) {
    if (N < 0 || T <= 0 || H <= 0 || W <= 0 || Cout <= 0) return STEP_E_SHAPE;
    if (!dw) return STEP_E_NULL;
    if (!accumulate) {
        const int e = (int)hipMemsetAsync(dw, 0, (size_t)Cout * 3 * 343 * sizeof(float), (hipStream_t)stream);
        if (e != 0) return e;
    }
    if (N == 0) return STEP_OK;
    if (!x || !dy) return STEP_E_NULL;
    StemWgradParams p;
    p.x = x; p.dy = dy; p.dw = dw; p.N = N; p.T = T; p.H = H; p.W = W;
    p.To = (T + 5 - 7) / 2 + 1; p.Ho = (H + 5 - 7) / 2 + 1; p.Wo = (W + 5 - 7) / 2 + 1;
    if (p.To <= 0 || p.Ho <= 0 || p.Wo <= 0) return STEP_E_SHAPE;
    p.Cout = Cout; p.cot = ceil_div(Cout, 64);
    p.rows = 8; p.hchunks = ceil_div(p.Ho, p.rows);       // 8 output rows per wavefront job: enough jobs for one clip
    p.jobs = (long long)N * p.To * p.hchunks;
    dim3 grid((unsigned)ceil_div64(p.jobs, 4), (unsigned)(49 * p.cot));
    switch (dtype) {
        case STEP_F32: STEP_LAUNCH((stem_wgrad_kernel<float>), grid, dim3(256), stream, p); break;
        case STEP_BF16: STEP_LAUNCH((stem_wgrad_kernel<bf16_t>), grid, dim3(256), stream, p); break;
        case STEP_F16: STEP_LAUNCH((stem_wgrad_kernel<f16_t>), grid, dim3(256), stream, p); break;
        default: return STEP_E_DTYPE;
    }
    return STEP_LAUNCH_CHECK();
}


}  // extern "C"
