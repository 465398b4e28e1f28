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
#include "conv_pw_kernel.h"
#include "pool_vec.h"
#include "options.h"
#include <stdlib.h>

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


// Separable TF-SAME max pool for the window / stride combinations of the backbone:
//   (3,3,3)/(1,1,1)  the branch_3 pool of every Inception block (models/i3dpt.py:151-155)
//   (1,3,3)/(1,2,2)  maxPool3d_2a_3x3, maxPool3d_3a_3x3      (models/i3dpt.py:193,199)
//   (3,3,3)/(2,2,2)  maxPool3d_4a_3x3                        (models/i3dpt.py:206)
// HBM-bound work: 2 bytes moved per element against up to 26 comparisons, so the job is to touch every input
// element once and to keep the comparisons cheap.
// A 256-thread workgroup owns a tile of TH x TW output pixels x 64 bytes of channels and walks along D.
// The 3-D max is separable and done in three stages per input plane:
//   W : the plane's input tile (LDS, (TH-1)*SH+KH rows x (TW-1)*SW+KW cols) -> row-wise KW-max at stride SW -> 2nd LDS buffer
//   H : column-wise KH-max at stride SH of that buffer -> the plane's 2-D window max, in registers
//   D : max with the previous planes' 2-D maxima, which every thread keeps in registers (rolling window);
//       an output plane is emitted every SD input planes
// i.e. KW + KH LDS reads per 16-byte 2-D result instead of KD*KH*KW global reads per output, every input plane is
// read from global memory once per column (the next plane's loads fly during the two passes), and all
// comparisons run in the storage type (VecMax).  Padding follows PoolParams: a position inside the explicit TF
// pad carries 0, a position beyond it (ceil-mode overhang) carries the lowest key and never wins.
// Tiles: at most 14x14 outputs at stride 1 and 7x7 at stride 2 (a 16x16 / 15x15 input tile); 28x28 maps split
// into 2x2 stride-1 tiles whose halos are L2 hits (the launch order keeps neighbouring tiles on one XCD).
constexpr int PP_SL = 4;                            // 16-byte vectors per pixel per workgroup (64 B contiguous)
constexpr int PP_R = 4;                             // load items per thread per plane: 16*16*4 / 256
constexpr int PP_MAXIN = 16;                        // max input tile edge

// (body as a device function: `bid` of `nblk` workgroups -- pool333_pw_kernel carries these workgroups in front of a pointwise conv's)
// (EXT: the two LDS images live in the caller's arena -- PP_LDS_BYTES, 16-byte aligned -- see conv_pw_body)
constexpr int PP_LDS_BYTES = 2 * PP_MAXIN * PP_MAXIN * PP_SL * 16;
template <typename T, int KD, int KH, int KW, int SD, int SH, int SW, int NT, bool EXT = false>
__device__ __forceinline__ void maxpool_sep_body(const T* __restrict__ x, T* __restrict__ y, const PoolParams& p,
                                                 int TH, int TW, int tiles_h, int tiles_w, int cchunks, int dseg, int nseg, int bid, int nblk,
                                                 unsigned char* arena = nullptr) {
    constexpr int V = elem<T>::VEC;
    typedef typename Vec16<T, V>::raw raw;
    static_assert(sizeof(raw) == 16, "16-byte vectors");
    constexpr int SL = PP_SL, R = PP_R * 256 / NT;        // NT = 256: 4 items per thread per pass, NT = 1024: 1
    raw *lds_raw, *lds_w;
    if constexpr (EXT) {
        lds_raw = (raw*)arena;
        lds_w = (raw*)arena + PP_MAXIN * PP_MAXIN * SL;
    } else {
        __shared__ __attribute__((aligned(16))) raw own_raw[PP_MAXIN * PP_MAXIN * SL];
        __shared__ __attribute__((aligned(16))) raw own_w[PP_MAXIN * PP_MAXIN * SL];
        lds_raw = own_raw; lds_w = own_w;
    }

    const int tid = threadIdx.x;
    // launch order -> XCD: consecutive workgroup ids go round-robin over the 8 XCDs; remap so that ids that
    // are neighbours in (tile, D segment) order -- which share halos -- land on the same XCD (one L2)
    int t = bid;
    if ((nblk & 7) == 0) t = (t & 7) * (nblk >> 3) + (t >> 3);
    // channel chunk fastest: a chunk is 64 bytes -- half a 128-byte line -- so the two chunks of a line are neighbours in launch
    // order on ONE XCD (the second one's loads hit that L2) instead of being fetched by two XCDs (measured: maxPool3d_2a 64 -> 45 us)
    const int cc = t % cchunks; t /= cchunks;
    const int tw_i = t % tiles_w; t /= tiles_w;
    const int th_i = t % tiles_h; t /= tiles_h;
    const int seg = t % nseg;
    const int n = t / nseg;
    const int oh0 = th_i * TH, ow0 = tw_i * TW;
    const int c0 = cc * SL * V;
    const int IR = (TH - 1) * SH + KH, IC = (TW - 1) * SW + KW;      // input tile

    raw zero;
#pragma unroll
    for (int i = 0; i < V; ++i) zero[i] = 0;
    const raw kzero = VecMax<T>::enc(zero), klow = VecMax<T>::lowest();

    // ---- per-thread item tables (the same every plane)
    int ld_goff[R], ld_loff[R];          // load items: element offset inside a plane (-1: pad = 0, -2: overhang = lowest), LDS index (-1: none)
    int wp_off[R];                       // W-pass items: lds_raw index of the window's first vector (lds_w index = the item itself)
    const int n_ld = IR * IC * SL, n_wp = IR * TW * SL, n_hp = TH * TW * SL;
#pragma unroll
    for (int q = 0; q < R; ++q) {
        const int item = tid + q * NT;
        {
            const int sl = item % SL, pix = item / SL;
            const int r = pix / IC, cl = pix % IC;
            const int pr = oh0 * SH + r, pc = ow0 * SW + cl;           // padded coordinates
            const int ih = pr - p.pfh, iw = pc - p.pfw, c = c0 + sl * V;
            ld_loff[q] = item < n_ld ? item : -1;
            ld_goff[q] = (pr >= p.Lph || pc >= p.Lpw) ? -2
                       : ((ih >= 0 && ih < p.H && iw >= 0 && iw < p.W && c < p.C) ? (ih * p.W + iw) * p.x_cstride + p.x_coff + c : -1);
        }
        {
            const int sl = item % SL, pix = item / SL;
            const int r = pix / TW, ow = pix % TW;
            wp_off[q] = item < n_wp ? (r * IC + ow * SW) * SL + sl : -1;
        }
    }
    // H-pass items: lds_w index of the window's first vector, output offset inside a plane (-1: none);
    // TH*TW*SL <= 784 items at stride 1 (4 per thread), <= 196 at stride 2 (1 per thread)
    constexpr int RH = (SH == 1 && SW == 1) ? R : 1;   // (NT = 1024: 1)
    int hp_offs[RH], hp_goffs[RH];
#pragma unroll
    for (int q = 0; q < RH; ++q) {
        const int item = tid + q * NT;
        const int sl = item % SL, pix = item / SL;
        const int oh = pix / TW, ow = pix % TW;
        const int gh = oh0 + oh, gw = ow0 + ow, c = c0 + sl * V;
        hp_offs[q] = item < n_hp ? (oh * SH * TW + ow) * SL + sl : -1;
        hp_goffs[q] = (item < n_hp && gh < p.Ho && gw < p.Wo && c < p.C) ? (gh * p.Wo + gw) * p.y_cstride + p.y_coff + c : -1;
    }
    const size_t xplane = (size_t)p.H * p.W * p.x_cstride, yplane = (size_t)p.Ho * p.Wo * p.y_cstride;
    const T* xn = x + (size_t)n * p.D * xplane;
    T* yn = y + (size_t)n * p.Do * yplane;

    const int pend_ = (min(seg * dseg + dseg, p.Do) - 1) * SD + KD;      // one past this column's last padded plane
    // padded plane pd <-> input plane pd - pfd; class 0: real, 1: explicit pad (all zero), 2: beyond the pad (never wins)
    auto plane_class = [&](int pd) { return pd >= p.Lpd ? 2 : ((pd - p.pfd >= 0 && pd - p.pfd < p.D) ? 0 : 1); };
    auto load_plane = [&](int pd, raw (&r)[R]) {
        const int d = pd - p.pfd;
        const bool real = plane_class(pd) == 0 && pd < pend_;
#pragma unroll
        for (int q = 0; q < R; ++q) {
            raw val = zero;
            if (real && ld_goff[q] >= 0) val = *(const raw*)(xn + (size_t)d * xplane + ld_goff[q]);
            r[q] = val;
        }
    };

    // output planes [obeg, oend) of this column <- padded input planes [obeg*SD, (oend-1)*SD + KD)
    const int obeg = seg * dseg, oend = min(obeg + dseg, p.Do);
    const int pbeg = obeg * SD, pend = (oend - 1) * SD + KD;
    // pad / overhang positions of the input tile are the same for every plane: written once
#pragma unroll
    for (int q = 0; q < R; ++q)
        if (ld_loff[q] >= 0 && ld_goff[q] < 0) lds_raw[ld_loff[q]] = ld_goff[q] == -2 ? klow : kzero;

    raw m1[RH], m2[RH];                                  // 2-D maxima of the two previous planes
#pragma unroll
    for (int q = 0; q < RH; ++q) { m1[q] = klow; m2[q] = klow; }
    // one input plane: rg holds it on entry and the next plane on exit (its loads fly during the two passes;
    // a second plane in flight measured no faster and costs 16 VGPRs)
    auto step = [&](int pd, raw (&rg)[R]) {
        const int cls = plane_class(pd);                 // workgroup-uniform
        raw m0[RH];
        if (cls == 0) {
#pragma unroll
            for (int q = 0; q < R; ++q)
                if (ld_goff[q] >= 0) lds_raw[ld_loff[q]] = VecMax<T>::enc(rg[q]);
            __syncthreads();                             // input tile visible; previous H pass done with lds_w
            load_plane(pd + 1, rg);
#pragma unroll
            for (int q = 0; q < R; ++q)
                if (wp_off[q] >= 0) {
                    const raw* s0 = lds_raw + wp_off[q];
                    raw m = s0[0];
#pragma unroll
                    for (int k = 1; k < KW; ++k) m = VecMax<T>::max(m, s0[k * SL]);
                    lds_w[tid + q * NT] = m;
                }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < RH; ++q) {
                m0[q] = klow;
                if (hp_offs[q] >= 0) {
                    const raw* s0 = lds_w + hp_offs[q];
                    raw m = s0[0];
#pragma unroll
                    for (int k = 1; k < KH; ++k) m = VecMax<T>::max(m, s0[k * TW * SL]);
                    m0[q] = m;
                }
            }
        } else {
            load_plane(pd + 1, rg);
#pragma unroll
            for (int q = 0; q < RH; ++q) m0[q] = cls == 1 ? kzero : klow;
        }
        // D stage: the window of output plane od = (pd - KD + 1) / SD ends at this plane
        const int w0p = pd - (KD - 1);
        const bool emit = w0p >= pbeg && (w0p % SD) == 0;
#pragma unroll
        for (int q = 0; q < RH; ++q) {
            if (emit && hp_goffs[q] >= 0) {
                raw m = m0[q];
                if (KD >= 2) m = VecMax<T>::max(m, m1[q]);
                if (KD >= 3) m = VecMax<T>::max(m, m2[q]);
                *(raw*)(yn + (size_t)(w0p / SD) * yplane + hp_goffs[q]) = VecMax<T>::dec(m);
            }
            m2[q] = m1[q]; m1[q] = m0[q];
        }
    };
    raw rg[R];
    load_plane(pbeg, rg);
    for (int pd = pbeg; pd < pend; ++pd) step(pd, rg);
}

template <typename T, int KD, int KH, int KW, int SD, int SH, int SW, int NT>
__global__ __launch_bounds__(NT) void maxpool_sep_kernel(const T* __restrict__ x, T* __restrict__ y, PoolParams p,
                                                          int TH, int TW, int tiles_h, int tiles_w, int cchunks, int dseg, int nseg) {
    maxpool_sep_body<T, KD, KH, KW, SD, SH, SW, NT>(x, y, p, TH, TW, tiles_h, tiles_w, cchunks, dseg, nseg, (int)blockIdx.x, (int)gridDim.x);
}

// The 3x3x3 / 1 pool of an Inception block AND the block's fused 1x1x1 convs in ONE grid: both read the block input, neither fills
// the chip on the 14x14 maps (the pool is 512 latency-bound workgroups, the convs 245-490) and one used to wait for the other.  The
// pool's workgroups come first, the 256-thread workgroups of conv_pw_body<T, 1, 4> behind them (cp.gbase = npool).  Occupancy: the two
// bodies' LDS images share ONE 43 KB arena (as static arrays of the two functions they added up to 75 KB) and the pointwise body runs
// a two-deep global -> register ring here (154 instead of 182 VGPRs): THREE workgroups of either kind per CU instead of two --
// C2 +0.4-1.0 %, C3 +1-2 % (same box, libraries swapped; the ring depth changes no arithmetic).
// NBC = 2 (round 6): the pointwise workgroups own 128 pixels x 128 channels (54 KB of LDS: two workgroups per CU) -- on MANY rows (the 25x25
// maps of 4 AVA clips, the heads' 7x7 maps of 4 x 34 tubes: 45-60 k rows, where both halves fill the chip several times over) the NB = 1 form's
// 64-channel workgroups re-read the input once per 64 output channels and run the GEMM at two thirds of the NB = 2 rate
// (profiles/r06_ab_pws_heads.txt: 832 -> 256 on 60 k rows 58.8 against 41.9 us stand-alone).
template <typename T, int NBC = 1>
__global__ __launch_bounds__(256, NBC == 1 ? 3 : 2) void pool333_pw_kernel(const T* __restrict__ x, T* __restrict__ y, PoolParams p, int TH, int TW, int tiles_h,
                                                                           int tiles_w, int cchunks, int dseg, int nseg, int npool, ConvParams cp) {
    constexpr int ARENA = PP_LDS_BYTES > conv_pw_lds_bytes<T, NBC, 4>() ? PP_LDS_BYTES : conv_pw_lds_bytes<T, NBC, 4>();
    __shared__ __attribute__((aligned(16))) unsigned char arena[ARENA];     // ONE arena for whichever body this workgroup runs
    if ((int)blockIdx.x < npool) maxpool_sep_body<T, 3, 3, 3, 1, 1, 1, 256, true>(x, y, p, TH, TW, tiles_h, tiles_w, cchunks, dseg, nseg, (int)blockIdx.x, npool, arena);
    else conv_pw_body<T, NBC, 4, true, 2>(cp, arena);
}

// Backward of the TF-SAME max pool (training): the gradient of an output goes to the FIRST maximum of its window in
// (d, h, w) scan order -- torch's MaxPool3d rule (`val > max`), applied to the explicitly zero-padded tensor as
// MaxPool3dTFPadding builds it (models/i3dpt.py:114-126): when a pad element wins, the gradient is dropped.
// One thread per (output pixel, channel), lanes along C (coalesced taps); fp32 atomics into gx.
template <typename T>
__global__ void maxpool3d_tf_bwd_kernel(const T* __restrict__ x, const float* __restrict__ gy, float* __restrict__ gx, PoolParams p,
                                        long long total) {
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)blockDim.x * gridDim.x) {
        const int c = (int)(idx % p.C);
        long long pix = idx / p.C;
        const int ow = (int)(pix % p.Wo); pix /= p.Wo;
        const int oh = (int)(pix % p.Ho); pix /= p.Ho;
        const int od = (int)(pix % p.Do);
        const int n = (int)(pix / p.Do);
        float best = -__builtin_inff();
        long long arg = -1;
        for (int a = 0; a < p.kd; ++a) {
            const int pd = od * p.sd + a;
            if (pd >= p.Lpd) break;
            const int id = pd - p.pfd;
            for (int b = 0; b < p.kh; ++b) {
                const int phh = oh * p.sh + b;
                if (phh >= p.Lph) break;
                const int ih = phh - p.pfh;
                for (int cc = 0; cc < p.kw; ++cc) {
                    const int pw = ow * p.sw + cc;
                    if (pw >= p.Lpw) break;
                    const int iw = pw - p.pfw;
                    const bool inb = id >= 0 && id < p.D && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
                    const long long pos = (((long long)n * p.D + id) * p.H + ih) * p.W + iw;
                    const float v = inb ? elem<T>::to_f32(x[(size_t)pos * p.x_cstride + p.x_coff + c]) : 0.f;
                    if (v > best) { best = v; arg = inb ? pos : -1; }
                }
            }
        }
        if (arg >= 0) atomicAdd(gx + (size_t)arg * p.C + c, gy[idx]);
    }
}

// ---- max-pool backward as two gathers (no atomics, no clear, no fp32 staging) ------------------------------------------
// maxpool3d_tf_bwd_kernel above scatters with fp32 atomics into a cleared fp32 tensor that the caller then converts: for a
// 16-bit net ~24 bytes of traffic per input element and 2.3 ms of the 20 ms C4 step.  Same rule, gathered:
//   pass A (maxpool_arg_kernel): one BYTE per output element = the tap (scan order (a * kh + b) * kw + c) of the FIRST maximum
//     of its zero-padded window -- torch's `val > max` walk; a padding tap can win, and then no input owns the gradient;
//   pass B (maxpool_bwd_gather_kernel): every input element visits the <= kd*kh*kw / (sd*sh*sw) windows that contain it and
//     adds the gradients of those whose winning tap is its own position, in a fixed (od, oh, ow) order, and writes the result
//     once in the activation type: ~4.5 bytes per element, bit-reproducible.
// Lanes run along 16-byte channel vectors of x (8 channels for the 16-bit types, 4 for fp32).
template <typename TG, int V>
__device__ __forceinline__ void load_vec_f32(const TG* p, float (&f)[V]) {
    if constexpr (sizeof(TG) == 4) {
#pragma unroll
        for (int q = 0; q < V / 4; ++q) {
            const f32x4 v = *(const f32x4*)((const float*)p + 4 * q);
            f[4 * q] = v[0]; f[4 * q + 1] = v[1]; f[4 * q + 2] = v[2]; f[4 * q + 3] = v[3];
        }
    } else {
        static_assert(V == 8, "16-bit vectors carry 8 elements");
        Vec16<TG, 8>::unpack(*(const typename Vec16<TG, 8>::raw*)p, f);
    }
}
template <typename TO, int V>
__device__ __forceinline__ void store_vec_f32(TO* p, const float (&f)[V]) {
    if constexpr (sizeof(TO) == 4) {
#pragma unroll
        for (int q = 0; q < V / 4; ++q) {
            const f32x4 v = {f[4 * q], f[4 * q + 1], f[4 * q + 2], f[4 * q + 3]};
            *(f32x4*)((float*)p + 4 * q) = v;
        }
    } else {
        static_assert(V == 8, "16-bit vectors carry 8 elements");
        *(typename Vec16<TO, 8>::raw*)p = Vec16<TO, 8>::pack(f);
    }
}

// (Measured and removed, round 4: an XCD-aware launch order for the two gather passes -- every XCD a contiguous eighth of each
// grid-stride round, so that the workgroups sharing rows and planes share one L2 -- was SLOWER on the large maps (8 AVA clips: 3b
// 0.87 -> 1.04 ms, 3c 1.15 -> 1.29 ms, the heads' 7x7 maps 0.55 -> 0.77 ms per backward; 25x25 maps -5 %): the round-robin order
// spreads the simultaneous requests over all L2 channels, the contiguous order concentrates them.  tools/pool_bwd_bench.py.)
template <typename T>
__global__ void maxpool_arg_kernel(const T* __restrict__ x, unsigned char* __restrict__ arg, PoolParams p, long long total) {
    constexpr int V = elem<T>::VEC;
    typedef typename Vec16<T, V>::raw raw;
    const int CV = p.C / V;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)blockDim.x * gridDim.x) {
        const int cv = (int)(idx % CV);
        long long pix = idx / CV;
        const int ow = (int)(pix % p.Wo); pix /= p.Wo;
        const int oh = (int)(pix % p.Ho); pix /= p.Ho;
        const int od = (int)(pix % p.Do);
        const int n = (int)(pix / p.Do);
        float best[V];
        unsigned char win[V];
#pragma unroll
        for (int i = 0; i < V; ++i) { best[i] = -__builtin_inff(); win[i] = 255; }    // 255: no tap compared greater (all NaN / -inf): nobody owns the gradient, as in the atomic form
        for (int a = 0; a < p.kd; ++a) {
            const int pd = od * p.sd + a;
            if (pd >= p.Lpd) break;
            const int id = pd - p.pfd;
            for (int b = 0; b < p.kh; ++b) {
                const int phh = oh * p.sh + b;
                if (phh >= p.Lph) break;
                const int ih = phh - p.pfh;
                for (int c = 0; c < p.kw; ++c) {
                    const int pw = ow * p.sw + c;
                    if (pw >= p.Lpw) break;
                    const int iw = pw - p.pfw;
                    float f[V];
#pragma unroll
                    for (int i = 0; i < V; ++i) f[i] = 0.f;             // explicit TF padding: value 0
                    if (id >= 0 && id < p.D && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W)
                        Vec16<T, V>::unpack(*(const raw*)(x + ((((size_t)n * p.D + id) * p.H + ih) * p.W + iw) * p.x_cstride + p.x_coff + cv * V), f);
                    const unsigned char t = (unsigned char)((a * p.kh + b) * p.kw + c);
#pragma unroll
                    for (int i = 0; i < V; ++i)
                        if (f[i] > best[i]) { best[i] = f[i]; win[i] = t; }
                }
            }
        }
        unsigned char* dst = arg + (size_t)idx * V;
#pragma unroll
        for (int i = 0; i < V; ++i) dst[i] = win[i];
    }
}

template <typename T, typename TG, typename TO>
__global__ void maxpool_bwd_gather_kernel(const unsigned char* __restrict__ arg, const TG* __restrict__ gy, TO* __restrict__ gx, PoolParams p,
                                          long long total) {
    constexpr int V = elem<T>::VEC;
    const int CV = p.C / V;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)blockDim.x * gridDim.x) {
        const int cv = (int)(idx % CV);
        long long pix = idx / CV;
        const int iw = (int)(pix % p.W); pix /= p.W;
        const int ih = (int)(pix % p.H); pix /= p.H;
        const int id = (int)(pix % p.D);
        const int n = (int)(pix / p.D);
        const int pd = id + p.pfd, ph = ih + p.pfh, pw = iw + p.pfw;       // padded coordinates
        float sum[V];
#pragma unroll
        for (int i = 0; i < V; ++i) sum[i] = 0.f;
        // windows that contain the position: o*s <= q <= o*s + k - 1
        const int od0 = max(0, (pd - p.kd + p.sd) / p.sd), od1 = min(p.Do - 1, pd / p.sd);
        const int oh0 = max(0, (ph - p.kh + p.sh) / p.sh), oh1 = min(p.Ho - 1, ph / p.sh);
        const int ow0 = max(0, (pw - p.kw + p.sw) / p.sw), ow1 = min(p.Wo - 1, pw / p.sw);
        for (int od = od0; od <= od1; ++od)
            for (int oh = oh0; oh <= oh1; ++oh)
                for (int ow = ow0; ow <= ow1; ++ow) {
                    const unsigned char t = (unsigned char)(((pd - od * p.sd) * p.kh + (ph - oh * p.sh)) * p.kw + (pw - ow * p.sw));
                    const size_t o = ((((size_t)n * p.Do + od) * p.Ho + oh) * p.Wo + ow) * p.C + cv * V;
                    const unsigned char* wv = arg + o;
                    float g[V];
                    load_vec_f32<TG, V>(gy + o, g);
#pragma unroll
                    for (int i = 0; i < V; ++i)
                        if (wv[i] == t) sum[i] += g[i];
                }
        store_vec_f32<TO, V>(gx + (size_t)idx * V, sum);
    }
}

// ---- (3,3,3) / (1,1,1) max-pool backward in ONE launch (the branch_3 pool of every Inception block, i3dpt.py:151-155) ------------
// The two gathers above walk 27 taps per output and 27 windows per input through L1: 0.8 TB/s of operand traffic, 7 of the 55 ms
// of a C4 step at 8 clips x 15 tubes.  The first maximum of a zero-padded 3x3x3 window in scan order is SEPARABLE -- the first
// plane whose plane maximum is the window maximum, in it the first row whose row maximum is, in it the first column -- and so is
// the route of the gradient back: an output hands its gradient to a plane (3 candidates), the plane position to a row (3), the row
// position to a column (3).  Nine compares forth and nine back instead of 27 + 27, on tiles held in LDS, every x / gy element read
// from memory once (plus the tile halo) and no byte map in memory.
// A 512-thread workgroup owns a TH x TW tile of the map x CVC 16-byte channel vectors and walks the planes of one sample; a thread
// owns one position of the (TH+2) x (TW+2) grid of outputs that touch the tile and keeps its three-plane windows (plane maxima,
// winning planes, gradients) in registers.  Per input plane t, four barriers:
//   X tile (TH+4 x TW+4, zero outside the map) -> LDS | rows: first maximum of 3 columns -> (value, c) | columns: first maximum of
//   3 rows -> plane maximum M_t (registers) and b | depth: a(t-1) from M_{t-2..t} | plane q = t-2: G1 = sum over the 3 outputs of
//   the column whose a points at q -> LDS | G2[row] = sum over the 3 positions whose b points at the row -> LDS | gx[row][col] =
//   sum over the 3 positions whose c points at the column -> memory.
// Sums are fp32 in this fixed nested order (planes, then rows, then columns, each ascending); the gather form adds the same terms
// in (od, oh, ow) order, so the two agree to fp32 rounding, not bit for bit (torch's own backward adds them with atomics).
// "No winner" (every tap NaN or -inf) is tag 3 at each level and owns nothing, like the byte map's 255.
constexpr int PB_THREADS = 512;
constexpr int PB_XSLOTS = 768;          // (TH+4) x (TW+4) x CVC 16-byte slots of the input tile

template <typename T> __device__ __forceinline__ unsigned short exact_bits16(float f);      // f is representable in T: no rounding involved
template <> __device__ __forceinline__ unsigned short exact_bits16<bf16_t>(float f) { return (unsigned short)(__builtin_bit_cast(unsigned, f) >> 16); }
template <> __device__ __forceinline__ unsigned short exact_bits16<f16_t>(float f) { return elem<f16_t>::bits16(f); }

// first maximum of three 16-byte vectors in order: `val > best` from -inf, as torch walks a window (NaN never wins); tag 3: none
template <typename T>
__device__ __forceinline__ void first_max3(const u16x8& v0, const u16x8& v1, const u16x8& v2, float (&best)[8], unsigned (&win)[8]) {
    float f[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { best[i] = -__builtin_inff(); win[i] = 3; }
    Vec16<T, 8>::unpack(v0, f);
#pragma unroll
    for (int i = 0; i < 8; ++i) if (f[i] > best[i]) { best[i] = f[i]; win[i] = 0; }
    Vec16<T, 8>::unpack(v1, f);
#pragma unroll
    for (int i = 0; i < 8; ++i) if (f[i] > best[i]) { best[i] = f[i]; win[i] = 1; }
    Vec16<T, 8>::unpack(v2, f);
#pragma unroll
    for (int i = 0; i < 8; ++i) if (f[i] > best[i]) { best[i] = f[i]; win[i] = 2; }
}
__device__ __forceinline__ unsigned long long pack_tags(const unsigned (&w)[8]) {
    const unsigned lo = w[0] | (w[1] << 8) | (w[2] << 16) | (w[3] << 24), hi = w[4] | (w[5] << 8) | (w[6] << 16) | (w[7] << 24);
    return (unsigned long long)lo | ((unsigned long long)hi << 32);
}
__device__ __forceinline__ unsigned tag_of(unsigned long long w, int i) { return (unsigned)(w >> (8 * i)) & 0xffu; }

template <typename T, typename TG, typename TO>
__global__ __launch_bounds__(PB_THREADS) void maxpool333_bwd_kernel(const T* __restrict__ x, const TG* __restrict__ gy, TO* __restrict__ gx, PoolParams p,
                                                                    int TH, int TW, int tiles_h, int tiles_w, int CVC, int cchunks) {
    static_assert(sizeof(T) == 2, "16-bit activations");
    typedef u16x8 raw;
    // XS [0, 12 K) and RM [16 K, 28 K) live during the forward half of a plane step, G1S [0, 16 K) and G2S [16 K, 32 K) during the
    // backward half; each buffer's last reader is a barrier ahead of the next writer of its bytes (see the phase list above)
    __shared__ __attribute__((aligned(16))) unsigned char U[32768];
    __shared__ unsigned long long CR[3][PB_THREADS], BR[3][PB_THREADS];
    raw* const XS = (raw*)U;
    raw* const RM = (raw*)(U + 16384);
    float* const G1S = (float*)U;
    float* const G2S = (float*)(U + 16384);
    const int tid = threadIdx.x;
    const int CV = p.C / 8;
    long long bid = blockIdx.x;
    const int cc = (int)(bid % cchunks); bid /= cchunks;
    const int twi = (int)(bid % tiles_w); bid /= tiles_w;
    const int thi = (int)(bid % tiles_h);
    const int n = (int)(bid / tiles_h);
    const int h0 = thi * TH, w0 = twi * TW, cv0 = cc * CVC;
    const int PW = TW + 2, PH = TH + 2, XW = TW + 4, XH = TH + 4;
    const int nX = XH * XW * CVC, nR = XH * PW * CVC, nP = PH * PW * CVC;
    // input-tile items (two per thread): offset inside a plane, or -1 outside the map (zero padding)
    long long xoff[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int i = tid + k * PB_THREADS;
        xoff[k] = -1;
        if (i < nX) {
            const int cvi = i % CVC, pos = i / CVC, h = h0 - 2 + pos / XW, w = w0 - 2 + pos % XW;
            if (h >= 0 && h < p.H && w >= 0 && w < p.W && cv0 + cvi < CV) xoff[k] = ((long long)h * p.W + w) * p.x_cstride + p.x_coff + (cv0 + cvi) * 8;
        }
    }
    // this thread's output position (ph, pw) of the (TH+2) x (TW+2) grid, and the tile position it writes
    const bool pthread = tid < nP;
    const int pcvi = tid % CVC, ppos = tid / CVC, ph = ppos / PW, pw = ppos % PW;
    long long goff = -1, ooff = -1;
    if (pthread && cv0 + pcvi < CV) {
        const int oh = h0 - 1 + ph, ow = w0 - 1 + pw;
        if (oh >= 0 && oh < p.H && ow >= 0 && ow < p.W) goff = ((long long)oh * p.W + ow) * p.C + (cv0 + pcvi) * 8;
        if (ph < TH && pw < TW && h0 + ph < p.H && w0 + pw < p.W) ooff = ((long long)(h0 + ph) * p.W + (w0 + pw)) * p.C + (cv0 + pcvi) * 8;
    }
    const bool g2thread = pthread && ph < TH, g3thread = g2thread && pw < TW;
    const long long xplane = (long long)p.H * p.W * p.x_cstride, gplane = (long long)p.H * p.W * p.C;
    const T* const xn = x + (long long)n * p.D * xplane;
    const TG* const gyn = gy + (long long)n * p.D * gplane;
    TO* const gxn = gx + (long long)n * p.D * gplane;
    const raw zero = {0, 0, 0, 0, 0, 0, 0, 0};
    raw xpre[2] = {zero, zero};
#pragma unroll
    for (int k = 0; k < 2; ++k)
        if (xoff[k] >= 0) xpre[k] = *(const raw*)(xn + xoff[k]);                 // plane 0
    raw M0 = zero, M1 = zero, M2 = zero;                                           // plane maxima of planes t-2, t-1, t (plane -1: the zero pad)
    unsigned long long a0 = 0x0303030303030303ull, a1 = a0, a2 = a0;               // winning planes of outputs t-3, t-2, t-1 (3: none / no such output)
    float g0[8], g1[8], g2[8], gpre[8];                                            // gradients of outputs t-3, t-2, t-1; the next one in flight
#pragma unroll
    for (int i = 0; i < 8; ++i) g0[i] = g1[i] = g2[i] = gpre[i] = 0.f;
    if (goff >= 0) load_vec_f32<TG, 8>(gyn + goff, gpre);                          // output plane 0
    for (int t = 0; t <= p.D + 1; ++t) {
        M0 = M1; M1 = M2;
        if (t < p.D) {
            // ---- forward half: plane t -> row maxima -> plane maximum of this thread's position
#pragma unroll
            for (int k = 0; k < 2; ++k)
                if (tid + k * PB_THREADS < nX) XS[tid + k * PB_THREADS] = xpre[k];
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                xpre[k] = zero;
                if (t + 1 < p.D && xoff[k] >= 0) xpre[k] = *(const raw*)(xn + (long long)(t + 1) * xplane + xoff[k]);      // next plane: in flight during the passes
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int i = tid + k * PB_THREADS;
                if (i < nR) {
                    const int cvi = i % CVC, pos = i / CVC, xh = pos / PW, rw = pos % PW;
                    const raw* src = XS + (xh * XW + rw) * CVC + cvi;
                    float best[8]; unsigned win[8];
                    first_max3<T>(src[0], src[CVC], src[2 * CVC], best, win);
                    raw m;
#pragma unroll
                    for (int e = 0; e < 8; ++e) m[e] = exact_bits16<T>(best[e]);
                    RM[i] = m;
                    if (xh >= 2 && xh < TH + 2) CR[t % 3][((xh - 2) * PW + rw) * CVC + cvi] = pack_tags(win);
                }
            }
            __syncthreads();
            if (pthread) {
                const raw* src = RM + (ph * PW + pw) * CVC + pcvi;
                float best[8]; unsigned win[8];
                first_max3<T>(src[0], src[PW * CVC], src[2 * PW * CVC], best, win);
#pragma unroll
                for (int e = 0; e < 8; ++e) M2[e] = exact_bits16<T>(best[e]);
                BR[t % 3][tid] = pack_tags(win);
            }
        } else {
            M2 = zero;                                                             // planes D, D + 1: the zero pad behind the sample
        }
        // ---- depth: the winning plane of output t - 1 (planes t-2, t-1, t)
        a0 = a1; a1 = a2; a2 = 0x0303030303030303ull;
#pragma unroll
        for (int i = 0; i < 8; ++i) { g0[i] = g1[i]; g1[i] = g2[i]; g2[i] = 0.f; }
        if (t >= 1 && t - 1 < p.D) {
            float best[8]; unsigned win[8];
            first_max3<T>(M0, M1, M2, best, win);
            a2 = pack_tags(win);
#pragma unroll
            for (int i = 0; i < 8; ++i) g2[i] = gpre[i];
#pragma unroll
            for (int i = 0; i < 8; ++i) gpre[i] = 0.f;
            if (t < p.D && goff >= 0) load_vec_f32<TG, 8>(gyn + (long long)t * gplane + goff, gpre);       // output plane t, used in the next step
        }
        const int q = t - 2;
        if (q < 0 || q >= p.D) continue;                  // (uniform: every thread of the workgroup takes the same branch)
        // ---- backward half for input plane q: outputs q-1, q, q+1 own it through a = 2, 1, 0
        if (pthread) {
            float s[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                s[i] = tag_of(a0, i) == 2 ? g0[i] : 0.f;
                s[i] += tag_of(a1, i) == 1 ? g1[i] : 0.f;
                s[i] += tag_of(a2, i) == 0 ? g2[i] : 0.f;
            }
            *(f32x4*)(G1S + (size_t)tid * 4) = f32x4{s[0], s[1], s[2], s[3]};               // (low / high channel halves in separate planes: a lane's
            *(f32x4*)(G1S + PB_THREADS * 4 + (size_t)tid * 4) = f32x4{s[4], s[5], s[6], s[7]};   //  16 bytes next to its neighbour's, no bank conflicts)
        }
        __syncthreads();
        if (g2thread) {                                   // row ph of the tile: positions ph, ph+1, ph+2 of the output grid reach it through b = 2, 1, 0
            float s[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) s[i] = 0.f;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int o = tid + j * PW * CVC;
                const unsigned long long b = BR[q % 3][o];
                const f32x4 lo = *(const f32x4*)(G1S + (size_t)o * 4), hi = *(const f32x4*)(G1S + PB_THREADS * 4 + (size_t)o * 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    s[i] += tag_of(b, i) == (unsigned)(2 - j) ? lo[i] : 0.f;
                    s[i + 4] += tag_of(b, i + 4) == (unsigned)(2 - j) ? hi[i] : 0.f;
                }
            }
            *(f32x4*)(G2S + (size_t)tid * 4) = f32x4{s[0], s[1], s[2], s[3]};
            *(f32x4*)(G2S + PB_THREADS * 4 + (size_t)tid * 4) = f32x4{s[4], s[5], s[6], s[7]};
        }
        __syncthreads();
        if (g3thread && ooff >= 0) {                      // column pw: positions pw, pw+1, pw+2 of the row reach it through c = 2, 1, 0
            float s[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) s[i] = 0.f;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int o = tid + j * CVC;
                const unsigned long long c = CR[q % 3][o];
                const f32x4 lo = *(const f32x4*)(G2S + (size_t)o * 4), hi = *(const f32x4*)(G2S + PB_THREADS * 4 + (size_t)o * 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    s[i] += tag_of(c, i) == (unsigned)(2 - j) ? lo[i] : 0.f;
                    s[i + 4] += tag_of(c, i + 4) == (unsigned)(2 - j) ? hi[i] : 0.f;
                }
            }
            store_vec_f32<TO, 8>(gxn + (long long)q * gplane + ooff, s);
        }
        // (the next step's first barrier stands between these reads of G2S / CR and the row pass that overwrites RM;
        //  its XS writes overlap G1S, whose readers are behind the barrier above)
    }
}

// (Measured and NOT kept, round 4: the same separable, LDS-tiled treatment for the (1,3,3) / (1,2,2) pools (maxPool3d_2a / 3a: planes are
// independent, an input belongs to at most 2 x 2 windows).  Five barriers per 8 x 8-output tile with a third to a half of the 512 threads
// busy per phase: 2a 0.869 -> 0.888 ms, 3a 0.587 -> 0.727 ms at 8 AVA clips -- the two gathers already run these at 2.3-2.6 TB/s because
// every input visits only 4 windows.  tools/pool_bwd_bench.py.)
struct PoolBwdPlan { int TH, TW, tiles_h, tiles_w, CVC, cchunks; long long blocks; };
static PoolBwdPlan pool333_bwd_plan(const PoolParams& p) {
    PoolBwdPlan pl;
    const int CV = p.C / 8;
    auto shape = [&](int maxt) {
        pl.tiles_h = ceil_div(p.H, maxt); pl.tiles_w = ceil_div(p.W, maxt);
        pl.TH = ceil_div(p.H, pl.tiles_h); pl.TW = ceil_div(p.W, pl.tiles_w);
        int cvc = PB_THREADS / ((pl.TH + 2) * (pl.TW + 2));
        while (cvc > 1 && (pl.TH + 4) * (pl.TW + 4) * cvc > PB_XSLOTS) --cvc;
        if (cvc > 4) cvc = 4;                              // 64 contiguous bytes per pixel are enough; more chunks = more workgroups
        if (cvc > CV) cvc = CV;
        pl.CVC = cvc; pl.cchunks = ceil_div(CV, cvc);
        pl.blocks = (long long)p.N * pl.tiles_h * pl.tiles_w * pl.cchunks;
    };
    shape(14);
    if (pl.blocks < 512 && (p.H > 7 || p.W > 7)) shape(7);       // small batches: smaller tiles rather than idle CUs
    return pl;
}

template <typename T, typename TG, typename TO>
static void bwd333_launch(const void* x, const void* gy, void* gx, const PoolParams& p, const PoolBwdPlan& pl, step_stream_t stream) {
    STEP_LAUNCH((maxpool333_bwd_kernel<T, TG, TO>), dim3((unsigned)pl.blocks), dim3(PB_THREADS), stream, (const T*)x, (const TG*)gy, (TO*)gx, p,
                pl.TH, pl.TW, pl.tiles_h, pl.tiles_w, pl.CVC, pl.cchunks);
}

// Clip ingest: decoded frames arrive as uint8 [N,T,H,W,3] (what cv2 / the data loader hands over,
// data/ava.py:298-338); the reference converts on the host -- ConvertFromInts(scale), SubtractMeans, DivideStds
// (data/augmentations.py:68-111,600-612) -- and ships fp32 [T,3,H,W] over PCIe (4x the bytes).  Here the uint8 frames
// are transferred and this kernel writes the normalised clip in the layout BaseNet.forward takes ([N,T,3,H,W]).
// Same fp32 operation order as the numpy code (no FMA contraction): ((x*2)/255 - 1 - mean[c]) / std[c] for scale 2.
template <typename T>
__global__ void clip_from_u8_kernel(const unsigned char* __restrict__ src, T* __restrict__ dst, int HW, long long frames, int scale,
                                    float m0, float m1, float m2, float s0, float s1, float s2, long long total) {
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)blockDim.x * gridDim.x) {
        const long long f = idx / HW;
        const int pix = (int)(idx % HW);
        const unsigned char* s3 = src + (f * HW + pix) * 3;
        T* d = dst + f * 3 * HW + pix;
        const float mean[3] = {m0, m1, m2}, stdv[3] = {s0, s1, s2};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = (float)s3[c];
            if (scale == 1) v = __fdiv_rn(v, 255.f);
            else if (scale == 2) v = __fsub_rn(__fdiv_rn(__fmul_rn(v, 2.f), 255.f), 1.f);
            v = __fdiv_rn(__fsub_rn(v, mean[c]), stdv[c]);
            d[(size_t)c * HW] = elem<T>::from_f32(v);
        }
    }
}

// The same conversion, 16 consecutive pixels of one frame per thread (H*W % 16 == 0, 16-byte aligned buffers): three 16-byte loads of the
// interleaved bytes, per plane 16 * sizeof(T) contiguous bytes stored as 16-byte vectors -- a wave reads 3 KB and writes 1-2 KB per plane,
// all contiguous (the per-pixel form issues byte loads and 2-byte stores).  The same fp32 operations in the same order per element:
// bit-identical (kernel case).
template <typename T>
__global__ __launch_bounds__(256) void clip_from_u8_vec_kernel(const unsigned char* __restrict__ src, T* __restrict__ dst, int HW, int scale,
                                                               float m0, float m1, float m2, float s0, float s1, float s2, long long total16) {
    const int per_frame = HW >> 4;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total16; idx += (long long)blockDim.x * gridDim.x) {
        const long long f = idx / per_frame;
        const int pix = (int)(idx % per_frame) << 4;
        const u32x4* s4 = (const u32x4*)(src + (f * HW + pix) * 3);
        const u32x4 q0 = s4[0], q1 = s4[1], q2 = s4[2];
        const unsigned w[12] = {q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3], q2[0], q2[1], q2[2], q2[3]};
        const float mean[3] = {m0, m1, m2}, stdv[3] = {s0, s1, s2};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            T o[16];
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                const int b = 3 * p + c;
                float v = (float)((w[b >> 2] >> (8 * (b & 3))) & 0xffu);
                if (scale == 1) v = __fdiv_rn(v, 255.f);
                else if (scale == 2) v = __fsub_rn(__fdiv_rn(__fmul_rn(v, 2.f), 255.f), 1.f);
                v = __fdiv_rn(__fsub_rn(v, mean[c]), stdv[c]);
                o[p] = elem<T>::from_f32(v);
            }
            u32x4* d4 = (u32x4*)(dst + (f * 3 + c) * HW + pix);
#pragma unroll
            for (int v_ = 0; v_ < (int)sizeof(T); ++v_) {       // sizeof(T) 16-byte vectors per 16 elements
                u32x4 ov;
                __builtin_memcpy(&ov, (const char*)o + 16 * v_, 16);
                d4[v_] = ov;
            }
        }
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

static inline unsigned pool_flat_grid(long long total, int block) {
    long long g = ceil_div64(total, block);
    if (g > 16384) g = 16384;
    return (unsigned)g;
}

// tiling of the separable pool: balanced tiles of at most 14x14 outputs at stride 1 (7x7 at stride 2), 64 bytes of channels per
// workgroup, D cut into segments
struct PoolSepPlan { int TH, TW, tiles_h, tiles_w, cchunks, dseg, nseg; long long blocks; };
static PoolSepPlan pool_sep_plan(const PoolParams& p, int V) {
    PoolSepPlan sp;
    const int maxt = p.sh == 1 ? 14 : 7;
    sp.tiles_h = ceil_div(p.Ho, maxt); sp.tiles_w = ceil_div(p.Wo, maxt); sp.cchunks = ceil_div(p.C, PP_SL * V);
    sp.TH = ceil_div(p.Ho, sp.tiles_h); sp.TW = ceil_div(p.Wo, sp.tiles_w);
    sp.blocks = (long long)p.N * sp.tiles_h * sp.tiles_w * sp.cchunks;
    sp.dseg = p.Do; sp.nseg = 1;
    if (sp.blocks == 0) return sp;
    // split D until there are ~1000 workgroups; with kd = 3 every segment re-reads planes of its neighbours,
    // so keep >= 4 output planes per segment (kd = 1: no dependence along D)
    constexpr int target = 1024;
    // (single-tile maps -- the 14x14 stage -- are latency-bound chains of planes, not bandwidth-bound: shorter segments, measured
    // 14.8 -> 13.8 us per pool; on the 28x28 maps the extra halo planes cost more than the parallelism gives: 25.9 -> 30.2 us)
    const int minseg = p.kd == 1 ? 1 : (sp.tiles_h * sp.tiles_w == 1 ? 2 : 4);
    int nseg = 1;
    while (sp.blocks * nseg < target && p.Do / (nseg * 2) >= minseg) nseg *= 2;
    sp.dseg = ceil_div(p.Do, nseg);
    sp.nseg = ceil_div(p.Do, sp.dseg);
    return sp;
}

template <typename T>
static int maxpool_t(const void* x, void* y, const PoolParams& p, step_stream_t stream) {
    constexpr int V = elem<T>::VEC;
    if (p.C % V || p.x_cstride % V || p.x_coff % V || p.y_cstride % V || p.y_coff % V) return STEP_E_ALIGN;
    const int ksig = p.kd * 100 + p.kh * 10 + p.kw, ssig = p.sd * 100 + p.sh * 10 + p.sw;
    const bool sep = (ksig == 333 && ssig == 111) || (ksig == 133 && ssig == 122) || (ksig == 333 && ssig == 222);
    const bool force_direct = opt(STEP_OPT_POOL_DIRECT) != 0;        // tests: the general kernel on the separable shapes
    if (sep && !force_direct) {
        const PoolSepPlan sp = pool_sep_plan(p, V);
        if (sp.blocks == 0) return STEP_OK;
        const int TH = sp.TH, TW = sp.TW, tiles_h = sp.tiles_h, tiles_w = sp.tiles_w, cchunks = sp.cchunks, dseg = sp.dseg, nseg = sp.nseg;
        const dim3 grid((unsigned)(sp.blocks * nseg));
#define STEP_POOL_SEP(KD_, KH_, KW_, SD_, SH_, SW_) \
        STEP_LAUNCH((maxpool_sep_kernel<T, KD_, KH_, KW_, SD_, SH_, SW_, 256>), grid, dim3(256), stream, (const T*)x, (T*)y, p, TH, TW, tiles_h, tiles_w, cchunks, dseg, nseg)
        if (ssig == 111) { STEP_POOL_SEP(3, 3, 3, 1, 1, 1); }
        else if (ksig == 133) { STEP_POOL_SEP(1, 3, 3, 1, 2, 2); }
        else { STEP_POOL_SEP(3, 3, 3, 2, 2, 2); }
#undef STEP_POOL_SEP
        return STEP_LAUNCH_CHECK();
    }
    long long total = (long long)p.N * p.Do * p.Ho * p.Wo * (p.C / V);
    if (total == 0) return STEP_OK;
    STEP_LAUNCH((maxpool3d_tf_kernel<T>), dim3(pool_flat_grid(total, 256)), dim3(256), stream, (const T*)x, (T*)y, p, total);
    return STEP_LAUNCH_CHECK();
}

template <typename T>
static int avgpool_t(const void* x, void* y, int N, int D, int H, int W, int C, int kh, int kw, step_stream_t stream) {
    constexpr int V = elem<T>::VEC;
    if (C % V) return STEP_E_ALIGN;
    long long total = (long long)N * D * (H - kh + 1) * (W - kw + 1) * (C / V);
    if (total == 0) return STEP_OK;
    STEP_LAUNCH((avgpool_hw_kernel<T>), dim3(pool_flat_grid(total, 256)), dim3(256), stream, (const T*)x, (T*)y, N * D, H, W,
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

template <typename T, typename TG, typename TO>
static void bwd_gather_launch(const unsigned char* arg, const void* gy, void* gx, const PoolParams& p, long long total, step_stream_t stream) {
    STEP_LAUNCH((maxpool_bwd_gather_kernel<T, TG, TO>), dim3(pool_flat_grid(total, 256)), dim3(256), stream, arg, (const TG*)gy, (TO*)gx, p, total);
}
template <typename T>
static int bwd_gather_t(const void* x, int gy_dtype, const void* gy, int gx_dtype, void* gx, unsigned char* arg, const PoolParams& p,
                        step_stream_t stream) {
    constexpr int V = elem<T>::VEC;
    const bool gf = gy_dtype == STEP_F32, of = gx_dtype == STEP_F32;
    if constexpr (sizeof(T) == 2) {
        // the Inception blocks' (3,3,3) / (1,1,1) pool: one LDS-tiled launch, no byte map (STEP_OPT_POOL_DIRECT = 1 keeps the two gathers: tests)
        if (p.kd == 3 && p.kh == 3 && p.kw == 3 && p.sd == 1 && p.sh == 1 && p.sw == 1 && opt(STEP_OPT_POOL_DIRECT) == 0) {
            const PoolBwdPlan pl = pool333_bwd_plan(p);
            if (pl.blocks > 0 && pl.blocks <= 0x7fffffffLL && (pl.TH + 2) * (pl.TW + 2) * pl.CVC <= PB_THREADS && (pl.TH + 4) * (pl.TW + 4) * pl.CVC <= PB_XSLOTS) {
                if (gf && of) bwd333_launch<T, float, float>(x, gy, gx, p, pl, stream);
                else if (gf) bwd333_launch<T, float, T>(x, gy, gx, p, pl, stream);
                else if (of) bwd333_launch<T, T, float>(x, gy, gx, p, pl, stream);
                else bwd333_launch<T, T, T>(x, gy, gx, p, pl, stream);
                return STEP_LAUNCH_CHECK();
            }
        }
    }
    const long long touts = (long long)p.N * p.Do * p.Ho * p.Wo * (p.C / V), tins = (long long)p.N * p.D * p.H * p.W * (p.C / V);
    STEP_LAUNCH((maxpool_arg_kernel<T>), dim3(pool_flat_grid(touts, 256)), dim3(256), stream, (const T*)x, arg, p, touts);
    if (gf && of) bwd_gather_launch<T, float, float>(arg, gy, gx, p, tins, stream);
    else if (gf) bwd_gather_launch<T, float, T>(arg, gy, gx, p, tins, stream);
    else if (of) bwd_gather_launch<T, T, float>(arg, gy, gx, p, tins, stream);
    else bwd_gather_launch<T, T, T>(arg, gy, gx, p, tins, stream);
    return STEP_LAUNCH_CHECK();
}

// the combined launch behind step_pool_conv_forward (conv_igemm.hip prepares the conv's parameter block): 16-bit storage only
int pool333_pw_launch(int dtype, const void* x, int N, int D, int H, int W, int C, int x_cstride, int x_coff, void* y, int y_cstride, int y_coff,
                      ConvParams cp, long long conv_blocks, int nbc, step_stream_t stream) {
    if (dtype != STEP_BF16 && dtype != STEP_F16) return STEP_E_UNSUPPORTED;
    constexpr int V = 8;
    if (C % V || x_cstride % V || x_coff % V || y_cstride % V || y_coff % V) return STEP_E_UNSUPPORTED;
    PoolParams p;
    p.N = N; p.D = D; p.H = H; p.W = W; p.C = C; p.x_cstride = x_cstride; p.x_coff = x_coff;
    p.Do = pool_out_size(D, 3, 1); p.Ho = pool_out_size(H, 3, 1); p.Wo = pool_out_size(W, 3, 1);
    p.y_cstride = y_cstride; p.y_coff = y_coff;
    p.kd = p.kh = p.kw = 3; p.sd = p.sh = p.sw = 1;
    p.pfd = p.pfh = p.pfw = tf_pad_front(3, 1);
    p.Lpd = D + tf_pad_total(3, 1); p.Lph = H + tf_pad_total(3, 1); p.Lpw = W + tf_pad_total(3, 1);
    const PoolSepPlan sp = pool_sep_plan(p, V);
    const long long npool = sp.blocks * sp.nseg;
    if (npool <= 0 || npool + conv_blocks > 0x7fffffffLL) return STEP_E_UNSUPPORTED;
    cp.gbase = (int)npool; cp.gcount = (int)conv_blocks;
    const dim3 grid((unsigned)(npool + conv_blocks));
    if (dtype == STEP_BF16 && nbc == 2)
        STEP_LAUNCH((pool333_pw_kernel<bf16_t, 2>), grid, dim3(256), stream, (const bf16_t*)x, (bf16_t*)y, p, sp.TH, sp.TW, sp.tiles_h, sp.tiles_w, sp.cchunks,
                    sp.dseg, sp.nseg, (int)npool, cp);
    else if (dtype == STEP_BF16)
        STEP_LAUNCH((pool333_pw_kernel<bf16_t>), grid, dim3(256), stream, (const bf16_t*)x, (bf16_t*)y, p, sp.TH, sp.TW, sp.tiles_h, sp.tiles_w, sp.cchunks,
                    sp.dseg, sp.nseg, (int)npool, cp);
    else if (nbc == 2)
        STEP_LAUNCH((pool333_pw_kernel<f16_t, 2>), grid, dim3(256), stream, (const f16_t*)x, (f16_t*)y, p, sp.TH, sp.TW, sp.tiles_h, sp.tiles_w, sp.cchunks,
                    sp.dseg, sp.nseg, (int)npool, cp);
    else
        STEP_LAUNCH((pool333_pw_kernel<f16_t>), grid, dim3(256), stream, (const f16_t*)x, (f16_t*)y, p, sp.TH, sp.TW, sp.tiles_h, sp.tiles_w, sp.cchunks,
                    sp.dseg, sp.nseg, (int)npool, cp);
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

int step_maxpool3d_tf_backward(int dtype, const void* x, int N, int D, int H, int W, int C, int x_cstride, int x_coff, int kd,
                               int kh, int kw, int sd, int sh, int sw, const float* gy, float* gx, step_stream_t stream) {
    if (N < 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0 || kd <= 0 || kh <= 0 || kw <= 0 || sd <= 0 || sh <= 0 || sw <= 0)
        return STEP_E_SHAPE;
    if (x_coff < 0 || x_coff + C > x_cstride) return STEP_E_SHAPE;
    if (N == 0) return STEP_OK;
    if (!x || !gy || !gx) return STEP_E_NULL;
    PoolParams p;
    p.N = N; p.D = D; p.H = H; p.W = W; p.C = C; p.x_cstride = x_cstride; p.x_coff = x_coff;
    p.Do = pool_out_size(D, kd, sd); p.Ho = pool_out_size(H, kh, sh); p.Wo = pool_out_size(W, kw, sw);
    p.y_cstride = C; p.y_coff = 0;
    p.kd = kd; p.kh = kh; p.kw = kw; p.sd = sd; p.sh = sh; p.sw = sw;
    p.pfd = tf_pad_front(kd, sd); p.pfh = tf_pad_front(kh, sh); p.pfw = tf_pad_front(kw, sw);
    p.Lpd = D + tf_pad_total(kd, sd); p.Lph = H + tf_pad_total(kh, sh); p.Lpw = W + tf_pad_total(kw, sw);
    int rc = (int)hipMemsetAsync(gx, 0, sizeof(float) * (size_t)N * D * H * W * C, (hipStream_t)stream);
    if (rc != 0) return rc;
    const long long total = (long long)N * p.Do * p.Ho * p.Wo * C;
    const dim3 grid(pool_flat_grid(total, 256));
    switch (dtype) {
        case STEP_F32: STEP_LAUNCH((maxpool3d_tf_bwd_kernel<float>), grid, dim3(256), stream, (const float*)x, gy, gx, p, total); break;
        case STEP_BF16: STEP_LAUNCH((maxpool3d_tf_bwd_kernel<bf16_t>), grid, dim3(256), stream, (const bf16_t*)x, gy, gx, p, total); break;
        case STEP_F16: STEP_LAUNCH((maxpool3d_tf_bwd_kernel<f16_t>), grid, dim3(256), stream, (const f16_t*)x, gy, gx, p, total); break;
        default: return STEP_E_DTYPE;
    }
    return STEP_LAUNCH_CHECK();
}

int step_maxpool3d_tf_backward_gather(int dtype, const void* x, int N, int D, int H, int W, int C, int x_cstride, int x_coff, int kd, int kh,
                                      int kw, int sd, int sh, int sw, int gy_dtype, const void* gy, int gx_dtype, void* gx,
                                      unsigned char* arg_scratch, step_stream_t stream) {
    if (N < 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0 || kd <= 0 || kh <= 0 || kw <= 0 || sd <= 0 || sh <= 0 || sw <= 0)
        return STEP_E_SHAPE;
    if (x_coff < 0 || x_coff + C > x_cstride || kd * kh * kw > 255) return STEP_E_SHAPE;
    if (dtype != STEP_F32 && dtype != STEP_BF16 && dtype != STEP_F16) return STEP_E_DTYPE;
    if ((gy_dtype != STEP_F32 && gy_dtype != dtype) || (gx_dtype != STEP_F32 && gx_dtype != dtype)) return STEP_E_DTYPE;
    const int V = dtype == STEP_F32 ? 4 : 8;
    if (C % V || x_cstride % V || x_coff % V) return STEP_E_UNSUPPORTED;         // (the caller keeps step_maxpool3d_tf_backward)
    if (N == 0) return STEP_OK;
    if (!x || !gy || !gx || !arg_scratch) return STEP_E_NULL;
    if ((((uintptr_t)x) | ((uintptr_t)gy) | ((uintptr_t)gx)) & 15) return STEP_E_ALIGN;
    PoolParams p;
    p.N = N; p.D = D; p.H = H; p.W = W; p.C = C; p.x_cstride = x_cstride; p.x_coff = x_coff;
    p.Do = pool_out_size(D, kd, sd); p.Ho = pool_out_size(H, kh, sh); p.Wo = pool_out_size(W, kw, sw);
    p.y_cstride = C; p.y_coff = 0;
    p.kd = kd; p.kh = kh; p.kw = kw; p.sd = sd; p.sh = sh; p.sw = sw;
    p.pfd = tf_pad_front(kd, sd); p.pfh = tf_pad_front(kh, sh); p.pfw = tf_pad_front(kw, sw);
    p.Lpd = D + tf_pad_total(kd, sd); p.Lph = H + tf_pad_total(kh, sh); p.Lpw = W + tf_pad_total(kw, sw);
    switch (dtype) {
        case STEP_F32: return bwd_gather_t<float>(x, gy_dtype, gy, gx_dtype, gx, arg_scratch, p, stream);
        case STEP_BF16: return bwd_gather_t<bf16_t>(x, gy_dtype, gy, gx_dtype, gx, arg_scratch, p, stream);
        default: return bwd_gather_t<f16_t>(x, gy_dtype, gy, gx_dtype, gx, arg_scratch, p, stream);
    }
}

int step_clip_from_u8(const unsigned char* frames, int N, int T, int H, int W, int scale, const float* mean3, const float* std3,
                      int dtype, void* clip, step_stream_t stream) {
    if (N < 0 || T <= 0 || H <= 0 || W <= 0 || scale < 0 || scale > 2) return STEP_E_SHAPE;
    if (N == 0) return STEP_OK;
    if (!frames || !clip) return STEP_E_NULL;
    const float m0 = mean3 ? mean3[0] : 0.f, m1 = mean3 ? mean3[1] : 0.f, m2 = mean3 ? mean3[2] : 0.f;   // host pointers (3 floats)
    const float s0 = std3 ? std3[0] : 1.f, s1 = std3 ? std3[1] : 1.f, s2 = std3 ? std3[2] : 1.f;
    const long long fr = (long long)N * T, total = fr * H * W;
    if (dtype != STEP_F32 && dtype != STEP_BF16 && dtype != STEP_F16) return STEP_E_DTYPE;
    if ((H * W) % 16 == 0 && ((size_t)frames & 15) == 0 && ((size_t)clip & 15) == 0 && opt(STEP_OPT_CLIP_VEC) != 0) {
        const long long total16 = total / 16;
        const dim3 g16(pool_flat_grid(total16, 256));
        switch (dtype) {
            case STEP_F32: STEP_LAUNCH((clip_from_u8_vec_kernel<float>), g16, dim3(256), stream, frames, (float*)clip, H * W, scale, m0, m1, m2, s0, s1, s2, total16); break;
            case STEP_BF16: STEP_LAUNCH((clip_from_u8_vec_kernel<bf16_t>), g16, dim3(256), stream, frames, (bf16_t*)clip, H * W, scale, m0, m1, m2, s0, s1, s2, total16); break;
            default: STEP_LAUNCH((clip_from_u8_vec_kernel<f16_t>), g16, dim3(256), stream, frames, (f16_t*)clip, H * W, scale, m0, m1, m2, s0, s1, s2, total16); break;
        }
        return STEP_LAUNCH_CHECK();
    }
    const dim3 grid(pool_flat_grid(total, 256));
    switch (dtype) {
        case STEP_F32: STEP_LAUNCH((clip_from_u8_kernel<float>), grid, dim3(256), stream, frames, (float*)clip, H * W, fr, scale, m0, m1, m2, s0, s1, s2, total); break;
        case STEP_BF16: STEP_LAUNCH((clip_from_u8_kernel<bf16_t>), grid, dim3(256), stream, frames, (bf16_t*)clip, H * W, fr, scale, m0, m1, m2, s0, s1, s2, total); break;
        case STEP_F16: STEP_LAUNCH((clip_from_u8_kernel<f16_t>), grid, dim3(256), stream, frames, (f16_t*)clip, H * W, fr, scale, m0, m1, m2, s0, s1, s2, total); break;
        default: return STEP_E_DTYPE;
    }
    return STEP_LAUNCH_CHECK();
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
