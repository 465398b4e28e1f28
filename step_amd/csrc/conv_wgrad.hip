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
    float* ws;                    // partial tiles [blockIdx.x][blockIdx.y][MB*NB][16][64 lanes]: the four jobs of a workgroup already summed (NULL: fp32 atomics on dw), see wgrad_reduce_kernel
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
    // (the launch-order remap of conv_wgrad16_lds_kernel -- the groups of a pixel job back to back on one XCD -- was measured
    // here too and is NOT used: the jobs of a group end in fp32 atomics on the same dw tile, and bunching the groups changes which
    // atomics collide: fp32 C4 step 49.7 -> 60.4 ms, 5b_b1b weight gradient 0.27 -> 0.80 ms)
    const long long job = (long long)blockIdx.x * 4 + wave;
    const bool valid = job < p.jobs;                         // wave-uniform
    if (!valid && !p.ws) return;                             // (workspace form: the wave still takes part in the workgroup's sum, with zeros)
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

    const long long r_end = valid ? min((job + 1) * (long long)p.rows, p.total_rows) : 0;
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
    if (p.ws) {
        // The workgroup's four jobs are summed through LDS first -- (w0 + w2) + (w1 + w3), a fixed order -- and ONE partial tile per
        // workgroup goes to the workspace, accumulator layout as is (a register = 256 contiguous bytes over the lanes);
        // wgrad_reduce_kernel sums the workgroups of a tile in ascending order: no atomics (every job used to end in 4096 of them,
        // ~0.2 us per job at the rate the L2 sustains), bit-reproducible.  (One tile per JOB: 4x the workspace traffic -- with 128-pixel
        // jobs the 143 reduce launches of a C4 step read 4.9 ms worth of partial tiles.)
        constexpr int TILE = MB * NB * 16 * 64;
        __shared__ float red[2 * TILE];
        if (wave >= 2) {
            float* o = red + (wave - 2) * TILE + lane;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[((mb * NB + nb) * 16 + r) * 64] = acc[mb][nb][r];
        }
        __syncthreads();
        if (wave < 2) {
            const float* o = red + wave * TILE + lane;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mb][nb][r] += o[((mb * NB + nb) * 16 + r) * 64];
        }
        __syncthreads();
        if (wave == 1) {
            float* o = red + lane;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[((mb * NB + nb) * 16 + r) * 64] = acc[mb][nb][r];
        }
        __syncthreads();
        if (wave == 0) {
            const float* o = red + lane;
            float* out = p.ws + (((size_t)blockIdx.x * gridDim.y + blockIdx.y) * (MB * NB * 16)) * 64 + lane;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) out[((mb * NB + nb) * 16 + r) * 64] = acc[mb][nb][r] + o[((mb * NB + nb) * 16 + r) * 64];
        }
        return;
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

// sums the partial tiles of conv_wgrad_kernel over the jobs of the pixel axis: one thread per (tile, register, lane), jobs in
// ascending order; writes (or adds to) dw in torch's weight layout
__device__ __forceinline__ void wgrad_reduce_body(const float* __restrict__ ws, float* __restrict__ dw, long long jobs, int gy, int MB, int NB, int cot,
                                                  int cit, int Cout, int Cin, int ntaps, int accumulate, unsigned bx, unsigned nbx) {
    const int per_tile = MB * NB * 16 * 64;
    const long long per_job = (long long)gy * per_tile;
    // (256 threads = 64 consecutive elements x 4 subgroups of the jobs, partial sums added in subgroup order through LDS: see wgrad16_reduce_body)
    __shared__ float part1[256];
    const int ln = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long chunks = (per_job + 63) / 64;
    for (long long ch = bx; ch < chunks; ch += nbx) {
        const long long idx = ch * 64 + ln;
        float sum = 0.f;
        if (idx < per_job) {
            long long j = w;
            for (; j + 12 < jobs; j += 16) {
                const float v0 = ws[(size_t)j * per_job + idx], v1 = ws[(size_t)(j + 4) * per_job + idx];
                const float v2 = ws[(size_t)(j + 8) * per_job + idx], v3 = ws[(size_t)(j + 12) * per_job + idx];
                sum = (((sum + v0) + v1) + v2) + v3;
            }
            for (; j < jobs; j += 4) sum += ws[(size_t)j * per_job + idx];
        }
        part1[threadIdx.x] = sum;
        __syncthreads();
        const bool mine = w == 0 && idx < per_job;               // one unconditional barrier pair, the work predicated (no barrier inside a branch)
        if (mine) sum = ((sum + part1[64 + ln]) + part1[128 + ln]) + part1[192 + ln];
        __syncthreads();
        if (!mine) continue;
        int t = (int)(idx % per_tile);
        int y = (int)(idx / per_tile);
        const int lane = t & 63, r = (t >> 6) & 15, tile = t >> 10;
        const int nb = tile % NB, mb = tile / NB;
        const int cit_i = y % cit; y /= cit;
        const int cot_i = y % cot;
        const int tap = y / cot;
        const int co = cot_i * 32 * MB + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), ci = cit_i * 32 * NB + nb * 32 + (lane & 31);
        if (co < Cout && ci < Cin) {
            float* o = dw + ((size_t)co * Cin + ci) * ntaps + tap;
            *o = accumulate ? *o + sum : sum;
        }
    }
}
__global__ void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, long long jobs, int gy, int MB, int NB, int cot, int cit,
                                    int Cout, int Cin, int ntaps, int accumulate) {
    wgrad_reduce_body(ws, dw, jobs, gy, MB, NB, cot, cit, Cout, Cin, ntaps, accumulate, blockIdx.x, gridDim.x);
}


// conv_wgrad16_lds_kernel -- the 16-bit weight gradient as a tiled GEMM (kw = 3 windows: 3x3x3 and 1x3x3).
// The form above loads every MFMA operand element with its own 2- or 4-byte load and is bound by that (20-130 TFLOP/s).
// Here a 384-thread workgroup stages, per chunk of output rows, the dY tile [P pixels][64 co] and the X halo tile
// [(R+2) x (W+2) pixels][64 ci] of one input plane into LDS in their NATURAL pixel-major layout (16-byte vectors, pitch 160 B)
// and reads the MFMA operands with ds_read_b64_tr_b16: the hardware transpose hands every lane the 8 consecutive pixels (k)
// of one channel that v_mfma_f32_16x16x32 wants, and a tap shift is just another pixel address -- no transposed or shifted
// copies (common.h: lds_tr8).  Six waves = 3 filter rows (kh) x 2 halves of the 64 input channels; each accumulates the three
// kw taps of its row: 3 x (64 co x 32 ci) = 24 MFMAs per 32-pixel step from 20 transpose reads (NW = 6; the product form of the 3x3
// windows since round 4 is NW = 12: the co tile split over a second pair of waves, conv_wgrad16_lds12_kernel below).  A workgroup walks several
// (plane, row chunk) units with its accumulators in registers and ends in one pass of fp32 atomics.
constexpr int WG16_P = 224;              // output pixels per chunk (7 k32 steps)
constexpr int WG16_XP = 320;             // halo pixels per chunk
constexpr int WG16_XP_WIDE = 416;        // ... of the wide-map instantiation: 100-wide maps take two rows per chunk, 56-wide ones four (100 KB of LDS,
                                         // 14 staged vectors per thread: 5 % slower than the 320 form when both give the same rows, so chosen per layer)
constexpr int WG16_PW_P = 128;           // pointwise form: pixels per chunk
// LDS pitches and the pixel order of the transpose reads (round 4, from the SQ counters: 40-43 % of these kernels' LDS cycles were bank
// conflicts, profiles/r04_pmc_c4.txt).  A ds_read_b64_tr_b16 moves 8 bytes per lane: 32 lanes fill the 64 banks once, and a half-wave
// (two 16-lane groups) reads 8 pixel rows x 32 bytes.  With the natural order -- group g takes rows 8g .. 8g + 7, four per read -- the
// half-wave's rows are k .. k+3 and k+8 .. k+11, and rows 8 apart share banks at ANY 16-byte-aligned pitch.  The MFMA does not care
// in which order the 32 pixels of a step meet, as long as both operands use the same one: group g now takes rows 16 (g >> 1) + 4 (g & 1)
// + {0..3} (+8 for its second read), so a half-wave reads 8 CONSECUTIVE rows, and a pitch of 160 bytes modulo 256 (40 dwords: 40 k mod 64
// runs through all eight multiples of 8) puts them on disjoint banks.
constexpr int WG16_PW_XPITCH = 416;      // ... bytes per X pixel (192 channels x 2 B + 32 B)
constexpr int WG16_PITCH = 160;          // bytes per LDS pixel: 64 channels x 2 B + 32 B
struct Wgrad16Params {
    const void* x; const void* dy; float* dw;
    int N, D, H, W, Cin, Cout, kd;
    int x_cstride, x_coff, dy_cstride, dy_coff;
    int cot, cit;                 // 64-channel tiles along Cout / Cin
    int rows, cpp;                // output rows per chunk, chunks per plane
    int upj;                      // (plane, chunk) units per workgroup
    long long units;
    float* ws;                    // partial tiles [gridDim.x][gridDim.y][6 waves][24 tiles][64 lanes][4] (NULL: fp32 atomics on dw)
    unsigned wmagic, wmagic2;     // floor(k / W) = (k * wmagic) >> 22, floor(k / (W + 2)) = (k * wmagic2) >> 22 for k < 512
};

// KW = 1 (pointwise convs / Linear layers): no halo; the unit is a chunk of 128 consecutive pixels of the flattened N*D*H*W
// axis, the six waves take six 32-channel blocks of a 192-channel ci tile (X image pitch 400 B) and accumulate one tap.
template <typename T, int KW, bool PF, int XPMAX = WG16_XP, int NW = 6, bool DB = false>
__device__ __forceinline__ void conv_wgrad16_lds_body(const Wgrad16Params& p) {
    static_assert(sizeof(T) == 2, "16-bit storage");
    static_assert(KW == 3 || KW == 1, "3x3 windows or pointwise");
    constexpr bool PW = KW == 1;
    static_assert(NW == 6 || (NW == 12 && KW == 3), "six waves, or twelve for the 3x3 windows");
    constexpr int THREADS = NW * 64;
    constexpr int MA = NW == 12 ? 2 : 4;                      // 16-channel blocks of the co tile per wave (twelve waves: two halves of the 64)
    constexpr int PITCH = WG16_PITCH;
    constexpr int XPITCH = PW ? WG16_PW_XPITCH : WG16_PITCH;
    constexpr int CIT = PW ? 192 : 64;                        // input channels per workgroup
    constexpr int XCV = CIT / 8;                              // 16-byte vectors per X pixel
    constexpr int LDSB = PW ? WG16_PW_P * (PITCH + WG16_PW_XPITCH) : (WG16_P + XPMAX) * PITCH;    // 72 KiB (two workgroups per CU) | 85 / 100 KiB
    // DB: two images -- unit u + 1 is written while slower waves still multiply unit u, and one barrier per unit instead of two
    static_assert(!DB || PF, "the double-buffered form prefetches across units");     // (2 x 87 KB no longer fits the LDS at the 160-byte pitch: experiment builds only, with a smaller WG16_P)
    __shared__ __attribute__((aligned(16))) unsigned char lds[DB ? 2 * LDSB : LDSB];
    unsigned char* dyI = lds;
    unsigned char* xI = lds + (PW ? WG16_PW_P : WG16_P) * PITCH;
    const int tid = threadIdx.x, lane = tid & 63;
#ifdef STEP_EMUL
    const int wave = tid >> 6;
#else
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
    const int khw = PW ? 0 : wave % 3;                                          // filter row
    const int wrest = PW ? wave : wave / 3;
    const int cb = NW == 12 ? (wrest & 1) : wrest, coh = NW == 12 ? (wrest >> 1) : 0;   // 32-channel block of the ci tile; half of the co tile
    // launch order -> XCD: all gridDim.y channel / plane groups of a pixel unit read the SAME two images.  Dispatched in (x fastest)
    // order they are gridDim.x workgroups apart -- each fetches the images from HBM / the infinity cache again (9 x 92 MB for
    // conv3d_2c).  Remap so that the groups of one unit are consecutive on ONE XCD (ids go round-robin over the 8 XCDs) and the
    // second to last of them hit its L2.
    int bx = blockIdx.x, by = blockIdx.y;
    if ((gridDim.x & 7) == 0) {
        const long long L = (long long)blockIdx.x + (long long)gridDim.x * blockIdx.y;
        const int xcd = (int)(L & 7);
        const long long slot = L >> 3;
        by = (int)(slot % gridDim.y);
        bx = (int)(slot / gridDim.y) * 8 + xcd;
    }
    int t = by;
    const int cit_i = t % p.cit; t /= p.cit;
    const int cot_i = t % p.cot;
    const int kd_ = t / p.cot;
    const int co0 = cot_i * 64, ci0 = cit_i * CIT;
    const int W2 = p.W + 2;
    const int g = lane >> 4, rr = (lane & 15) >> 2, q = lane & 3;

    f32x4 acc[KW][MA][2];
#pragma unroll
    for (int s = 0; s < KW; ++s)
#pragma unroll
        for (int ma = 0; ma < MA; ++ma)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[s][ma][nb][r] = 0.f;

    const T* xg = (const T*)p.x;
    const T* dyg = (const T*)p.dy;
    const long long u_beg = (long long)bx * p.upj, u_end = min(u_beg + p.upj, p.units);
    // A unit's two images go global -> registers -> LDS; the NEXT unit's vectors are requested before this unit's matrix
    // work, so their round trip (2-3 us under load, against ~1 us of MFMAs per unit) flies under it.
    constexpr int NV = PW ? (WG16_PW_P * (8 + XCV) + THREADS - 1) / THREADS : ((WG16_P + XPMAX) * 8 + THREADS - 1) / THREADS;     // 16-byte vectors per thread per unit (11 | 12 / 14)
    u32x4 stg[NV];
    struct Unit { int n, d, id, r0, R, P, Ppad, XP; long long k0; bool live; };
    auto unit_of = [&](long long u) {
        Unit q;
        if (PW) {                                            // pixels [u * 128, +128) of the flat pixel axis
            const long long M = (long long)p.N * p.D * p.H * p.W, k0 = u * WG16_PW_P;
            q.n = q.d = q.id = q.R = 0; q.r0 = 0; q.live = true;
            q.k0 = k0;
            q.P = (int)min((long long)WG16_PW_P, M - k0); q.Ppad = (q.P + 31) & ~31; q.XP = q.P;
            return q;
        }
        q.k0 = 0;
        const long long plane = u / p.cpp;
        const int chunk = (int)(u % p.cpp);
        q.r0 = chunk * p.rows;
        q.R = min(p.rows, p.H - q.r0);
        q.n = (int)(plane / p.D); q.d = (int)(plane % p.D);
        q.id = q.d + kd_ - p.kd / 2;
        q.live = q.id >= 0 && q.id < p.D;                    // else: this plane of taps sees only zero padding
        q.P = q.R * p.W; q.Ppad = (q.P + 31) & ~31; q.XP = (q.R + 2) * W2;
        return q;
    };
    // (i0, i1: the slice of the thread's NV vectors this call moves -- the form without cross-unit prefetch stages a unit in two halves,
    // so that only half the staging registers are live at a time)
    // (staging order: the 16 lanes of a ds_write_b128 pass write rows r and r + 1, which share eight banks at this pitch; pairing rows r and
    //  r + 4 instead removes those conflicts (SQ_LDS_BANK_CONFLICT 40 -> 22-32 % of the LDS cycles with it, 40 -> ~35 % without) but measured
    //  SLOWER -- 5c_b1b 0.385 -> 0.416 ms, the heads' 1x3x3 convs 0.197 -> 0.213 -- and is not kept: the kernel is not bound by its LDS cycles)
    auto srow = [](int u) { return u >> 3; };
    auto prefetch = [&](const Unit& q, int i0, int i1) {
        const size_t gp0 = PW ? (size_t)q.k0 : (((size_t)q.n * p.D + q.d) * p.H + q.r0) * p.W;
        const size_t xp0 = ((size_t)q.n * p.D + q.id) * p.H;
#pragma unroll
        for (int i = i0; i < i1; ++i) {
            const int v = tid + i * THREADS;
            u32x4 val = {0u, 0u, 0u, 0u};
            if (v < q.Ppad * 8) {                            // dY [Ppad][64 co]: zero tail, zero past Cout
                const int k = srow(v), cv = v & 7;
                if (k < q.P && co0 + cv * 8 < p.Cout) val = *(const u32x4*)(dyg + (gp0 + k) * p.dy_cstride + p.dy_coff + co0 + cv * 8);
            } else if (PW) {                                 // X [P][192 ci]: zero past the chunk / past Cin
                const int w = v - q.Ppad * 8;
                const int px = w / XCV, cv = w % XCV;
                if (px < q.P && ci0 + cv * 8 < p.Cin) val = *(const u32x4*)(xg + (gp0 + px) * p.x_cstride + p.x_coff + ci0 + cv * 8);
            } else {                                         // X halo [(R+2)(W+2)][64 ci]: zero border, zero past Cin
                const int w = v - q.Ppad * 8;
                const int px = srow(w), cv = w & 7;
                const int r_ = (int)(((unsigned)px * p.wmagic2) >> 22), c_ = px - r_ * W2;
                const int ih = q.r0 + r_ - 1, iw = c_ - 1;
                if (px < q.XP && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W && ci0 + cv * 8 < p.Cin)
                    val = *(const u32x4*)(xg + ((xp0 + ih) * p.W + iw) * p.x_cstride + p.x_coff + ci0 + cv * 8);
            }
            stg[i] = val;
        }
    };
    auto to_lds = [&](const Unit& q, int i0, int i1) {
#pragma unroll
        for (int i = i0; i < i1; ++i) {
            const int v = tid + i * THREADS;
            if (v < q.Ppad * 8) *(u32x4*)(dyI + srow(v) * PITCH + (v & 7) * 16) = stg[i];
            else if (PW) {
                const int w = v - q.Ppad * 8;
                if (w < q.XP * XCV) *(u32x4*)(xI + (w / XCV) * XPITCH + (w % XCV) * 16) = stg[i];
            } else {
                const int w = v - q.Ppad * 8;                // (whole 8-row blocks: the rows past XP of the last one carry zeros)
                if (w < ((q.XP + 7) & ~7) * 8) *(u32x4*)(xI + srow(w) * PITCH + (w & 7) * 16) = stg[i];
            }
        }
    };
    Unit cur = unit_of(u_beg < u_end ? u_beg : 0);
    if (PF && u_beg < u_end && cur.live) prefetch(cur, 0, NV);
    for (long long u = u_beg; u < u_end; ++u) {
        if (!PF) {                                           // no cross-unit prefetch: the staging registers die before the matrix phase (fewer
            cur = unit_of(u);                                // VGPRs -> two resident workgroups per CU, whose staging and matrix phases interleave)
            if (cur.live) prefetch(cur, 0, KW == 3 ? NV / 2 : NV);
        }
        if (DB) {                                            // this unit's images: the last reader of them (unit u - 2) is a barrier behind
            dyI = lds + (size_t)((u - u_beg) & 1) * LDSB;
            xI = dyI + (PW ? WG16_PW_P : WG16_P) * PITCH;
        } else
            __syncthreads();                                 // every wave is done with the previous unit's images
        if (!PF && KW == 3) {
            if (cur.live) {
                to_lds(cur, 0, NV / 2);
                prefetch(cur, NV / 2, NV);
                to_lds(cur, NV / 2, NV);
            }
        } else if (cur.live) to_lds(cur, 0, NV);
        __syncthreads();
        const Unit done = cur;
        if (PF && u + 1 < u_end) {
            cur = unit_of(u + 1);
            if (cur.live) prefetch(cur, 0, NV);
        }
        if (!done.live) continue;                            // (workgroup-uniform)
        const int P = done.P, Ppad = done.Ppad;
        // ---- 32 output pixels per step
        for (int kb = 0; kb < Ppad; kb += 32) {
            const unsigned char* pa[2];
            const unsigned char* pb[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int k = kb + 16 * (g >> 1) + 4 * (g & 1) + 8 * h + rr;       // this lane's pixel of the transpose block (order: see WG16_PITCH)
                const int kc = min(k, P - 1);                // (past the chunk: dY is zero there, X only has to be finite)
                const int row = (int)(((unsigned)kc * p.wmagic) >> 22);
                pa[h] = dyI + k * PITCH + q * 8;
                pb[h] = PW ? xI + kc * XPITCH + cb * 64 + q * 8 : xI + (kc + 2 * row + khw * W2) * PITCH + cb * 64 + q * 8;
            }
            u16x8 a[MA];
#pragma unroll
            for (int ma = 0; ma < MA; ++ma) a[ma] = lds_tr8(pa[0] + (coh * MA + ma) * 32, pa[1] + (coh * MA + ma) * 32);
#pragma unroll
            for (int s = 0; s < KW; ++s)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    const u16x8 b = lds_tr8(pb[0] + s * XPITCH + nb * 32, pb[1] + s * XPITCH + nb * 32);
#pragma unroll
                    for (int ma = 0; ma < MA; ++ma) mma16_k32(a[ma], b, acc[s][ma][nb], T());
                }
        }
    }
    if (p.ws) {
        // partial tile of this workgroup, accumulator layout as is (16-byte stores, fully coalesced); wgrad16_reduce_kernel
        // sums over the workgroups of the pixel axis in a fixed order -- no atomics, deterministic
        f32x4* out = (f32x4*)p.ws + ((((size_t)bx * gridDim.y + by) * NW + wave) * (KW * MA * 2)) * 64 + lane;
#pragma unroll
        for (int s = 0; s < KW; ++s)
#pragma unroll
            for (int ma = 0; ma < MA; ++ma)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) out[((s * MA + ma) * 2 + nb) * 64] = acc[s][ma][nb];
        return;
    }
    const int ntaps = PW ? 1 : p.kd * 9;
#pragma unroll
    for (int s = 0; s < KW; ++s) {
        const int tap = PW ? 0 : (kd_ * 3 + khw) * 3 + s;
#pragma unroll
        for (int ma = 0; ma < MA; ++ma)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = co0 + (coh * MA + ma) * 16 + 4 * g + r, ci = ci0 + cb * 32 + nb * 16 + (lane & 15);
                    if (co < p.Cout && ci < p.Cin) atomicAdd(p.dw + ((size_t)co * p.Cin + ci) * ntaps + tap, acc[s][ma][nb][r]);
                }
    }
}

template <typename T, int KW>
__global__ __launch_bounds__(384) void conv_wgrad16_lds_kernel(Wgrad16Params p) { conv_wgrad16_lds_body<T, KW, true>(p); }
// twelve waves (3 filter rows x 2 halves of the ci tile x 2 halves of the co tile, 12 accumulator tiles each): three waves per SIMD
// instead of 1.5, every SIMD equally loaded
// (measured, bf16, 8 AVA clips, same process: 3c_b1b 0.883 -> 0.669 ms = 714 TFLOP/s, 3b_b1b 0.582 -> 0.454, 4f_b1b 0.335 -> 0.270, 5c on 1080
// 7x7 maps 0.468 -> 0.387, the heads' 1x3x3 convs 0.262 -> 0.196; 142 VGPRs, no spill: the product form of the 3x3 windows.  The six-wave
// kernels remain for experiment builds, -DWG16_NW6.)
template <typename T, int KW>
#ifdef WG16_DB
__global__ __launch_bounds__(768) STEP_WAVES_PER_SIMD(3) void conv_wgrad16_lds12_kernel(Wgrad16Params p) { conv_wgrad16_lds_body<T, KW, true, WG16_XP, 12, true>(p); }
#else
__global__ __launch_bounds__(768) STEP_WAVES_PER_SIMD(3) void conv_wgrad16_lds12_kernel(Wgrad16Params p) { conv_wgrad16_lds_body<T, KW, true, WG16_XP, 12>(p); }
#endif
template <typename T, int KW>
__global__ __launch_bounds__(768) STEP_WAVES_PER_SIMD(3) void conv_wgrad16_lds12_wide_kernel(Wgrad16Params p) { conv_wgrad16_lds_body<T, KW, true, WG16_XP_WIDE, 12>(p); }
template <typename T, int KW>
__global__ __launch_bounds__(384) void conv_wgrad16_lds_wide_kernel(Wgrad16Params p) { conv_wgrad16_lds_body<T, KW, true, WG16_XP_WIDE>(p); }
// The same without the cross-unit register prefetch, held to 168 VGPRs: TWO workgroups per CU, whose staging and matrix phases interleave
// (and 12 waves spread evenly over the 4 SIMDs instead of 6).  Measured (tools/wgrad_bench.py, bf16, 8 AVA clips, same process):
// pointwise 480 -> 304 on 25x25x9x8: 137 -> 96 us, 832 -> 624 on 1080 7x7 maps: 914 -> 572 us (140 VGPRs, no spill) -- the product
// form for KW = 1; the 3x3 windows spill at 168 VGPRs and measured 15-25 % SLOWER than the prefetching form even with the staging
// split in two halves (profiles/r04_wgrad16_variants.txt): they keep one prefetching workgroup per CU.
template <typename T, int KW>
__global__ __launch_bounds__(384) STEP_WAVES_PER_SIMD(3) void conv_wgrad16_lds2_kernel(Wgrad16Params p) { conv_wgrad16_lds_body<T, KW, false>(p); }


// conv_wgrad16_pws_kernel -- pointwise (1x1x1 / Linear) weight gradient as a PIXEL STREAM (round 4).
// dW[Cout, Cin] = dY[M, Cout]^T X[M, Cin] is a GEMM whose K axis is the pixel axis: M = 10^4..10^6, Cout x Cin small.  Most of these
// layers are HBM-bound (2 x 128 x 256 / ((128 + 256) x 2 B) = 85 FLOP per operand byte at the largest tile), so the job is to read
// every operand byte once at the memory rate: the forms above cut Cout x Cin into 64 x 192 (or 64 x 64) tiles, re-read both operands
// once per tile of the other and stage a unit at a time without overlap -- 1/3 to 1/5 of the copy rate on the backbone's layers
// (the vendor GEMM is 2-5x slower still on these shapes: tools/gemm_probe.py).
// Here a 512-thread workgroup owns up to 128 x 256 of dW (8 waves as 2 x 4, 64 x 64 accumulators each: 64 VGPRs) and walks a slice of
// the pixel axis in 32-pixel stages: two stages of global loads in flight in registers (48 B per thread each), a double-buffered LDS
// image (pixel-major; pitch and pixel order of the transpose reads: see WG16_PITCH), ONE barrier per
// stage, fragments by ds_read_b64_tr_b16, 16 MFMAs (16x16x32) per wave and stage.  Two workgroups per CU (<= 128 VGPRs, 2 x 66 KB).
// Slices end in a dense fp32 [Cout, Cin] image per slice (the reduce is a coalesced sum of images) or, without a workspace, in atomics.
constexpr int PWS_P = 32;
constexpr int PWS_APITCH = 416;          // 128 output channels x 2 B + 160 (pitch = 160 mod 256 and the pixel order of WG16_PITCH: conflict-free half-waves)
constexpr int PWS_BPITCH = 672;          // 256 input channels x 2 B + 160
constexpr int PWS_STAGE = PWS_P * (PWS_APITCH + PWS_BPITCH);
struct WgradPwsParams {
    const void* x; const void* dy; float* dw; float* ws;
    long long M, ppj;             // pixels; pixels per slice (a multiple of 32)
    int Cin, Cout, x_cstride, x_coff, dy_cstride, dy_coff;
    int cot, cit, co_t, ci_t;     // tiles along Cout / Cin and their extents (co_t: a multiple of 32 <= 128, ci_t: a multiple of 64 <= 256)
};

template <typename T>
__global__ __launch_bounds__(512) STEP_WAVES_PER_SIMD_MIN(4) void conv_wgrad16_pws_kernel(WgradPwsParams p) {
    static_assert(sizeof(T) == 2, "16-bit storage");
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * PWS_STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
#ifdef STEP_EMUL
    const int wave = tid >> 6;
#else
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
    const int wm = wave & 1, wn = wave >> 1;
    // launch order -> XCD as in conv_wgrad16_lds_kernel: the tiles of one pixel slice consecutive on ONE XCD (they read the same pixels)
    int bx = blockIdx.x, by = blockIdx.y;
    if ((gridDim.x & 7) == 0) {
        const long long L = (long long)blockIdx.x + (long long)gridDim.x * blockIdx.y;
        const int xcd = (int)(L & 7);
        const long long slot = L >> 3;
        by = (int)(slot % gridDim.y);
        bx = (int)(slot / gridDim.y) * 8 + xcd;
    }
    const int cit_i = by % p.cit, cot_i = by / p.cit;
    const int co0 = cot_i * p.co_t, ci0 = cit_i * p.ci_t;
    const int cow = p.co_t >> 1, ciw = p.ci_t >> 2;           // channels per wave along Cout / Cin (multiples of 16)
    const int nma = cow >> 4, nnb = ciw >> 4;                  // 16-channel blocks per wave (1..4 each)
    const int g = lane >> 4, rr = (lane & 15) >> 2, q = lane & 3;
    f32x4 acc[4][4];
#pragma unroll
    for (int ma = 0; ma < 4; ++ma)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) acc[ma][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    const long long k_beg = (long long)bx * p.ppj, k_end = min(k_beg + p.ppj, p.M);
    const int S = k_beg < k_end ? (int)((k_end - k_beg + PWS_P - 1) / PWS_P) : 0;
    // staging: thread -> (pixel, 16-byte vector) of dY (one vector per stage) and of X (two); 32-bit element offsets from the slice's
    // first pixel (a slice is at most 2^31 elements of either operand: the planner checks)
    const int len = (int)(k_end > k_beg ? k_end - k_beg : 0);
    const T* const ag = (const T*)p.dy + k_beg * p.dy_cstride + p.dy_coff + co0;
    const T* const bgp = (const T*)p.x + k_beg * p.x_cstride + p.x_coff + ci0;
    const int apx = tid >> 4, acv = tid & 15;
    const bool a_in = acv * 8 < p.co_t, a_ld = a_in && co0 + acv * 8 < p.Cout;
    const unsigned aoff = (unsigned)apx * (unsigned)p.dy_cstride + (unsigned)acv * 8u;
    const int bpx0 = tid >> 5, bcv = tid & 31;                 // second vector: 16 pixels further
    const bool b_in = bcv * 8 < p.ci_t, b_ld = b_in && ci0 + bcv * 8 < p.Cin;
    const unsigned boff = (unsigned)bpx0 * (unsigned)p.x_cstride + (unsigned)bcv * 8u;
    struct Stage { u32x4 a, b0, b1; };
    auto load = [&](int s) {
        Stage st;
        st.a = st.b0 = st.b1 = u32x4{0u, 0u, 0u, 0u};
        const int r0 = s * PWS_P;                              // first pixel of the stage, relative to the slice
        if (a_ld && r0 + apx < len) st.a = *(const u32x4*)(ag + ((unsigned)r0 * (unsigned)p.dy_cstride + aoff));
        if (b_ld && r0 + bpx0 < len) st.b0 = *(const u32x4*)(bgp + ((unsigned)r0 * (unsigned)p.x_cstride + boff));
        if (b_ld && r0 + bpx0 + 16 < len) st.b1 = *(const u32x4*)(bgp + ((unsigned)(r0 + 16) * (unsigned)p.x_cstride + boff));
        return st;
    };
    auto store = [&](const Stage& st, int buf) {
        unsigned char* A = lds + buf * PWS_STAGE;
        unsigned char* B = A + PWS_P * PWS_APITCH;
        if (a_in) *(u32x4*)(A + apx * PWS_APITCH + acv * 16) = st.a;
        if (b_in) {
            *(u32x4*)(B + bpx0 * PWS_BPITCH + bcv * 16) = st.b0;
            *(u32x4*)(B + (bpx0 + 16) * PWS_BPITCH + bcv * 16) = st.b1;
        }
    };
    auto compute = [&](int buf) {
        const unsigned char* A = lds + buf * PWS_STAGE + wm * cow * 2 + q * 8;
        const unsigned char* B = lds + buf * PWS_STAGE + PWS_P * PWS_APITCH + wn * ciw * 2 + q * 8;
        const int k0 = 16 * (g >> 1) + 4 * (g & 1) + rr, k1 = k0 + 8;      // this lane's pixels of the two transpose blocks
        // (always the full 4 x 4 blocks: a narrower tile's surplus blocks multiply whatever the LDS rows hold behind its channels into
        //  accumulators that are never written out -- branch-free, and these layers are bound by their operand traffic, not by this)
        u16x8 a[4];
#pragma unroll
        for (int ma = 0; ma < 4; ++ma) a[ma] = lds_tr8(A + k0 * PWS_APITCH + ma * 32, A + k1 * PWS_APITCH + ma * 32);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            const u16x8 b = lds_tr8(B + k0 * PWS_BPITCH + nb * 32, B + k1 * PWS_BPITCH + nb * 32);
#pragma unroll
            for (int ma = 0; ma < 4; ++ma) mma16_k32(a[ma], b, acc[ma][nb], T());
        }
    };
    Stage s0 = load(0), s1 = load(1);
    for (int s = 0; s < S; s += 2) {
        store(s0, 0);
        s0 = load(s + 2);
        __syncthreads();
        compute(0);
        if (s + 1 < S) {                                        // (workgroup-uniform)
            store(s1, 1);
            s1 = load(s + 3);
            __syncthreads();
            compute(1);
        }
    }
    float* const out = p.ws ? p.ws + (size_t)bx * p.Cout * p.Cin : p.dw;
#pragma unroll
    for (int ma = 0; ma < 4; ++ma)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
            if (ma < nma && nb < nnb) {
                const int ci = ci0 + wn * ciw + nb * 16 + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = co0 + wm * cow + ma * 16 + 4 * g + r;
                    if (co < p.Cout && ci < p.Cin) {
                        if (p.ws) out[(size_t)co * p.Cin + ci] = acc[ma][nb][r];
                        else atomicAdd(out + (size_t)co * p.Cin + ci, acc[ma][nb][r]);
                    }
                }
            }
}

// fixed-order sum of the slices' dense [Cout, Cin] images.  A small layer has few elements and hundreds of slices: one thread per element
// walking all slices is a chain of dependent-latency loads on a handful of workgroups (measured: 100 us of a 160 us call).  So a 256-thread
// workgroup takes 16 consecutive 16-byte vectors x 16 slice subgroups (subgroup g adds slices g, g + 16, ... in ascending order, four loads
// in flight), and the 16 partial sums are added in subgroup order through LDS: the same association on every run.
__device__ __forceinline__ void wgradpws_reduce_body(const float* __restrict__ ws, float* __restrict__ dw, long long slices, long long n4, int accumulate,
                                                     unsigned bx, unsigned nbx) {
    __shared__ f32x4 part[256];
    const int t = threadIdx.x, vi = t & 15, sg = t >> 4;
    const long long chunks = (n4 + 15) / 16;
    for (long long ch = bx; ch < chunks; ch += nbx) {
        const long long i = ch * 16 + vi;
        f32x4 sum = {0.f, 0.f, 0.f, 0.f};
        if (i < n4) {
            long long sl = sg;
            for (; sl + 48 < slices; sl += 64) {
                const f32x4 v0 = ((const f32x4*)ws)[sl * n4 + i], v1 = ((const f32x4*)ws)[(sl + 16) * n4 + i];
                const f32x4 v2 = ((const f32x4*)ws)[(sl + 32) * n4 + i], v3 = ((const f32x4*)ws)[(sl + 48) * n4 + i];
#pragma unroll
                for (int e = 0; e < 4; ++e) sum[e] = (((sum[e] + v0[e]) + v1[e]) + v2[e]) + v3[e];
            }
            for (; sl < slices; sl += 16) {
                const f32x4 v = ((const f32x4*)ws)[sl * n4 + i];
#pragma unroll
                for (int e = 0; e < 4; ++e) sum[e] += v[e];
            }
        }
        part[t] = sum;
        __syncthreads();
        if (sg == 0 && i < n4) {
            f32x4 tot = accumulate ? ((const f32x4*)dw)[i] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const f32x4 v = part[k * 16 + vi];
#pragma unroll
                for (int e = 0; e < 4; ++e) tot[e] += v[e];
            }
            ((f32x4*)dw)[i] = tot;
        }
        __syncthreads();
    }
}
__global__ void wgradpws_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, long long slices, long long n4, int accumulate) {
    wgradpws_reduce_body(ws, dw, slices, n4, accumulate, blockIdx.x, gridDim.x);
}

// sums the partial tiles of conv_wgrad16_lds_kernel over the pixel-axis workgroups: one thread per (tile, lane) 16-byte group
// (Measured and not kept, round 4: the sum organised by OUTPUT -- a workgroup per 16 co x 16 ci block assembling [co][ci][27 taps] in LDS and
// writing whole rows instead of 4-byte stores 108 bytes apart: 3c_b1b at one clip 0.131 -> 0.145 ms, 16 -> 32 channels 0.038 -> 0.082 ms,
// the one-clip step 13.4 -> 13.7 ms.  The scattered stores are merged by the L2; a block-per-workgroup walk has too few workgroups.)
__device__ __forceinline__ void wgrad16_reduce_body(const float* __restrict__ ws, float* __restrict__ dw, int gx, int gy, int cot, int cit, int Cout,
                                                    int Cin, int kd, int accumulate, int pw, unsigned bx, unsigned nbx) {
    // pw: 0 = 3x3 windows, six waves x 24 tiles | 1 = pointwise, six waves x 8 tiles | 2 = 3x3 windows, twelve waves x 12 tiles
    const int tpw = pw == 1 ? 8 : (pw == 2 ? 12 : 24);                  // accumulator tiles per wave
    const int nw = pw == 2 ? 12 : 6;
    const long long per_x = (long long)gy * nw * tpw * 64;             // f32x4 groups of one pixel-axis workgroup
    const int ntaps = pw == 1 ? 1 : kd * 9;
    // 256 threads = 64 consecutive 16-byte groups x 4 subgroups of the pixel-axis workgroups (subgroup w adds x = w, w + 4, ... in ascending
    // order, four loads in flight; the four partial sums are added in subgroup order through LDS): one thread walking all gx partial tiles
    // was a chain of dependent-latency loads -- 60 us per grouped launch at one clip per GPU
    __shared__ f32x4 part[256];
    const int ln = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long chunks = (per_x + 63) / 64;
    for (long long ch = bx; ch < chunks; ch += nbx) {
        const long long idx = ch * 64 + ln;
        f32x4 sum = {0.f, 0.f, 0.f, 0.f};
        if (idx < per_x) {
            int x = w;
            for (; x + 12 < gx; x += 16) {
                const f32x4 v0 = ((const f32x4*)ws)[(size_t)x * per_x + idx], v1 = ((const f32x4*)ws)[(size_t)(x + 4) * per_x + idx];
                const f32x4 v2 = ((const f32x4*)ws)[(size_t)(x + 8) * per_x + idx], v3 = ((const f32x4*)ws)[(size_t)(x + 12) * per_x + idx];
#pragma unroll
                for (int e = 0; e < 4; ++e) sum[e] = (((sum[e] + v0[e]) + v1[e]) + v2[e]) + v3[e];
            }
            for (; x < gx; x += 4) {
                const f32x4 v = ((const f32x4*)ws)[(size_t)x * per_x + idx];
#pragma unroll
                for (int e = 0; e < 4; ++e) sum[e] += v[e];
            }
        }
        part[threadIdx.x] = sum;
        __syncthreads();
        if (w == 0 && idx < per_x) {
#pragma unroll
            for (int k = 1; k < 4; ++k) {
                const f32x4 v = part[k * 64 + ln];
#pragma unroll
                for (int e = 0; e < 4; ++e) sum[e] += v[e];
            }
            long long t = idx;
            const int lane = (int)(t % 64); t /= 64;
            const int tile = (int)(t % tpw); t /= tpw;
            const int wave = (int)(t % nw); t /= nw;
            const int y = (int)t;
            const int nb = tile & 1;
            const int khw = pw == 1 ? 0 : wave % 3, wrest = pw == 1 ? wave : wave / 3;
            const int cb = pw == 2 ? (wrest & 1) : wrest;
            const int ma = pw == 2 ? (wrest >> 1) * 2 + ((tile >> 1) & 1) : (tile >> 1) & 3, s = pw == 2 ? tile >> 2 : tile >> 3;
            int yy = y;
            const int cit_i = yy % cit; yy /= cit;
            const int cot_i = yy % cot;
            const int kd_ = yy / cot;
            const int tap = pw == 1 ? 0 : (kd_ * 3 + khw) * 3 + s;
            const int ci = cit_i * (pw == 1 ? 192 : 64) + cb * 32 + nb * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = cot_i * 64 + ma * 16 + 4 * (lane >> 4) + r;
                if (co < Cout && ci < Cin) {
                    float* o = dw + ((size_t)co * Cin + ci) * ntaps + tap;
                    *o = accumulate ? *o + sum[r] : sum[r];
                }
            }
        }
        __syncthreads();
    }
}
__global__ void wgrad16_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int gx, int gy, int cot, int cit, int Cout, int Cin,
                                      int kd, int accumulate, int pw) {
    wgrad16_reduce_body(ws, dw, gx, gy, cot, cit, Cout, Cin, kd, accumulate, pw, blockIdx.x, gridDim.x);
}

// the fixed-order sums of several layers' partial tiles as ONE launch (step_wgrad_reduce_group): blockIdx.y = the layer
struct WgradReduceGroup { step_wgrad_reduce_item it[STEP_WGRAD_REDUCE_MAX]; };
__global__ __launch_bounds__(256) void wgrad_reduce_group_kernel(WgradReduceGroup g) {
    const step_wgrad_reduce_item& it = g.it[blockIdx.y];
    if (it.kind == 1) wgrad_reduce_body(it.ws, it.dw, it.jobs, it.gy, 2, it.nbw, it.cot, it.cit, it.Cout, it.Cin, it.taps, it.accumulate, blockIdx.x, gridDim.x);
    else if (it.kind == 2) wgrad16_reduce_body(it.ws, it.dw, (int)it.jobs, it.gy, it.cot, it.cit, it.Cout, it.Cin, it.taps, it.accumulate, it.pw, blockIdx.x, gridDim.x);
    else if (it.kind == 3) wgradpws_reduce_body(it.ws, it.dw, it.jobs, (long long)it.Cout * it.Cin / 4, it.accumulate, blockIdx.x, gridDim.x);
}

// stem_wgrad_kernel -- weight gradient of the 7x7x7 stride-2 stem (Cin = 3) from the clip in its own [N,T,3,H,W]
// layout.  Same scheme as conv_wgrad_kernel (fp32 MFMA, reduction over output pixels, lanes along channels), but
// with only 3 input channels the B operand's 32 columns are the (kw, c) pairs of one (kd, kh) row of the filter
// (21 of 32 used): one wavefront job = one output plane (n, od) x one (kd, kh) x 64 output channels.
struct StemWgradParams {
    const void* x; const float* dy; float* dw;
    float* ws;                    // non-null: every job writes its partial tile here (stem_wgrad_reduce_kernel sums them in job order)
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
    float* wst = p.ws ? p.ws + (((size_t)job * gridDim.y + blockIdx.y) * 32) * 64 + lane : nullptr;
    if (it < 0 || it >= p.T) {
        if (wst)                                                   // a job outside the clip still owns its (zero) partial tile
#pragma unroll
            for (int q = 0; q < 32; ++q) wst[q * 64] = 0.f;
        return;
    }
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
    if (wst) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) wst[(mb * 16 + r) * 64] = acc[mb][r];
        return;
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

// sums the partial tiles of stem_wgrad_kernel over its jobs in job order: one thread per (filter-row tile, accumulator register, lane)
__global__ void stem_wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, long long jobs, int gy, int cot, int Cout,
                                         int accumulate) {
    const long long per_job = (long long)gy * 32 * 64;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < per_job; idx += (long long)blockDim.x * gridDim.x) {
        const int lane = (int)(idx % 64), q = (int)((idx / 64) % 32);
        int t = (int)(idx / (64 * 32));
        const int nn = lane & 31;
        const int cot_i = t % cot; t /= cot;
        const int kh_ = t % 7, kd_ = t / 7;
        const int co = cot_i * 64 + (q >> 4) * 32 + cd_row(q & 15, lane);
        if (nn >= 21 || co >= Cout) continue;
        float sum = 0.f;
        for (long long j = 0; j < jobs; ++j) sum += ws[(size_t)j * per_job + idx];
        float* o = dw + ((((size_t)co * 3 + nn % 3) * 7 + kd_) * 7 + kh_) * 7 + nn / 3;
        *o = accumulate ? *o + sum : sum;
    }
}


// stem_wgrad16_kernel -- the stem's weight gradient on the 16-bit matrix instructions (16-bit clip, 16-bit activation gradient).
// stem_wgrad_kernel gives every (kd, kh) filter row its own jobs, so the gradient tensor is read 49 times (9 GB for one
// 36 x 400 x 400 clip: 2.4 ms); here a workgroup keeps ALL 49 x 21 filter columns of a 32-channel block in registers:
//   seven waves = the seven kd planes, each holding five accumulators of the 32 x 32 product (rows = output channels, columns =
//   the 147 (kh, kw, c) entries of its filter plane in blocks of 32: 92 % of the columns used, where one filter row per block
//   would use 21 of 32); the reduction runs over the output pixels of a tile of SW16_R rows x P columns, 16 per matrix instruction.
//   A operand (dy^T): the tile's gradient rows staged pixel-major in LDS, read with the transposing ds_read_b64_tr_b16.
//   B operand (strided patches of the clip): the input rows a tile touches (7 planes x 2 R + 5 rows x 3 channels) staged with
//   even and odd columns DE-INTERLEAVED, so that the eight pixels of a fragment -- input columns 2 p + kw - 2, stride 2 -- are
//   eight consecutive elements of one phase (five aligned dwords + a byte funnel shift, lds_read8_shifted).
//   The workgroup walks tiles (the next tile's loads are in flight under the current one's matrix work) and writes its partial
//   [7][5][32 x 32] tiles once; stem_wgrad16_reduce_kernel sums them over the workgroups in a fixed order (no atomics).
constexpr int SW16_R = 2, SW16_PMAX = 112, SW16_ROWS = 2 * SW16_R + 5, SW16_PH = SW16_PMAX + 8;
constexpr int SW16_XROW = 3 * 2 * SW16_PH;                       // elements of one staged input row: 3 channels x 2 phases
constexpr int SW16_XS = 7 * SW16_ROWS * SW16_XROW;               // elements of the staged clip tile
constexpr int SW16_NT = 448;
constexpr int SW16_NXV = (7 * SW16_ROWS * 3 * (SW16_PMAX / 4 + 2) + SW16_NT - 1) / SW16_NT;
constexpr int SW16_NGV = (SW16_R * SW16_PMAX * 4 + SW16_NT - 1) / SW16_NT;

struct StemWgrad16Params {
    const void* x; const void* dy; float* ws;
    int N, T, H, W, To, Ho, Wo, Cout;
    int P, tiles_h, tiles_w;
    long long tiles;
};

// Eight consecutive 16-bit elements starting at an ODD or even element of a 4-byte aligned LDS row: five aligned dwords and a
// per-lane byte funnel shift (sh = 0 or 2).  A ds_read_b128 at a 2- or 4-byte aligned address is legal on gfx950 but measures
// 6.6x slower than an aligned one (tools/ubench/lds_align.hip); 32-bit reads run at the full LDS rate.
__device__ __forceinline__ u16x8 lds_read8_shifted(const unsigned char* p4, unsigned sh) {
    const unsigned* q = (const unsigned*)p4;
    const unsigned d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3], d4 = q[4];
    typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
#ifndef STEP_EMUL
    const u32x4_ r = {__builtin_amdgcn_alignbyte(d1, d0, sh), __builtin_amdgcn_alignbyte(d2, d1, sh), __builtin_amdgcn_alignbyte(d3, d2, sh),
                      __builtin_amdgcn_alignbyte(d4, d3, sh)};
#else
    auto ab = [](unsigned hi, unsigned lo, unsigned s_) { return (unsigned)(((((unsigned long long)hi) << 32) | lo) >> (8 * s_)); };
    const u32x4_ r = {ab(d1, d0, sh), ab(d2, d1, sh), ab(d3, d2, sh), ab(d4, d3, sh)};
#endif
    return __builtin_bit_cast(u16x8, r);
}
// The staging item tables (13 + 2 vectors per thread: plane / row / channel / column of each) are the same for every tile, and
// the compiler would keep all of them in registers across the tile loop -- 256 VGPRs, single-buffered fragments, spills.  An
// empty asm that "modifies" the item index makes them per-tile temporaries (a few hundred integer instructions per tile).
#ifndef STEP_EMUL
#define STEP_OPAQUE_V(v) asm volatile("" : "+v"(v))
#else
#define STEP_OPAQUE_V(v) ((void)0)
#endif

template <typename T>
__global__ __launch_bounds__(SW16_NT) void stem_wgrad16_kernel(StemWgrad16Params p) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[SW16_XS * 2 + SW16_R * SW16_PMAX * 64];
    unsigned char* xs = lds;
    unsigned char* gs = lds + SW16_XS * 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;          // wave = kd
    const int khalf = lane >> 5;
    const int co0 = blockIdx.y * 32;
    // staging items are numbered for the widest tile (constant divisors); a narrower tile skips the items beyond its width
    constexpr int VPRM = SW16_PMAX / 4 + 2;
    constexpr int nxv = 7 * SW16_ROWS * 3 * VPRM, ngv = SW16_R * SW16_PMAX * 4;
    const int P = p.P, VPR = P / 4 + 2;
    const T* xg = (const T*)p.x;
    const T* dyg = (const T*)p.dy;

    f32x16 acc[5];
#pragma unroll
    for (int nb = 0; nb < 5; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;

    u16x8 xr[SW16_NXV], gr[SW16_NGV];
    const u16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    auto load_tile = [&](long long tile) {
        long long t = tile;                                                // row tiles fastest: a workgroup's consecutive tiles share 5 of 9 input rows (L2)
        const int th = (int)(t % p.tiles_h); t /= p.tiles_h;
        const int tw = (int)(t % p.tiles_w); t /= p.tiles_w;
        const int od = (int)(t % p.To);
        const int n = (int)(t / p.To);
        const int oh0 = th * SW16_R, pw0 = tw * P;
#pragma unroll
        for (int q = 0; q < SW16_NXV; ++q) {
            int v = tid + q * SW16_NT;
            STEP_OPAQUE_V(v);                                                // (see STEP_OPAQUE_V)
            u16x8 val = zero8;
            if (v < nxv) {
                const int j = v % VPRM;
                int t2 = v / VPRM;
                const int c = t2 % 3; t2 /= 3;
                const int r = t2 % SW16_ROWS, kd = t2 / SW16_ROWS;
                const int it = 2 * od + kd - 2, ih = 2 * oh0 - 2 + r, iw = 2 * pw0 - 8 + 8 * j;
                if (j < VPR && it >= 0 && it < p.T && ih >= 0 && ih < p.H && iw >= 0 && iw + 8 <= p.W)
                    val = *(const u16x8*)(xg + ((((size_t)n * p.T + it) * 3 + c) * p.H + ih) * p.W + iw);
            }
            xr[q] = val;
        }
#pragma unroll
        for (int q = 0; q < SW16_NGV; ++q) {
            int v = tid + q * SW16_NT;
            STEP_OPAQUE_V(v);                                                // (see STEP_OPAQUE_V)
            u16x8 val = zero8;
            if (v < ngv) {
                const int cq = v & 3, pix = v >> 2;
                const int pp = pix % SW16_PMAX;
                const int oh = oh0 + pix / SW16_PMAX, ow = pw0 + pp;
                if (pp < P && oh < p.Ho && ow < p.Wo && co0 + 8 * cq + 8 <= p.Cout)
                    val = *(const u16x8*)(dyg + ((((size_t)n * p.To + od) * p.Ho + oh) * p.Wo + ow) * p.Cout + co0 + 8 * cq);
            }
            gr[q] = val;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int q = 0; q < SW16_NXV; ++q) {
            int v = tid + q * SW16_NT;
            STEP_OPAQUE_V(v);                                                // (see STEP_OPAQUE_V)
            if (v < nxv) {
                const int j = v % VPRM, row = v / VPRM;                         // row = (kd * ROWS + r) * 3 + c
                unsigned char* d = xs + ((size_t)row * 2 * SW16_PH + 4 * j) * 2;
                const u16x4 ev = {xr[q][0], xr[q][2], xr[q][4], xr[q][6]}, odd = {xr[q][1], xr[q][3], xr[q][5], xr[q][7]};
                *(u16x4*)d = ev;
                *(u16x4*)(d + SW16_PH * 2) = odd;
            }
        }
#pragma unroll
        for (int q = 0; q < SW16_NGV; ++q) {
            int v = tid + q * SW16_NT;
            STEP_OPAQUE_V(v);                                                // (see STEP_OPAQUE_V)
            if (v < ngv) *(u16x8*)(gs + (size_t)(v >> 2) * 64 + (v & 3) * 16) = gr[q];
        }
    };
    // this lane's fragment addresses.  B: column n = 32 nb + (lane & 31) = (kh, kw, c) of filter plane kd; input column
    // 2 (pw0 + p) + kw - 2 = staged element 2 p + kw + 6 -> phase kw & 1, index p + (kw >> 1) + 3; input row 2 ro + kh.
    // (surplus columns n >= 147 compute garbage that is never stored.)  A: common.h lds_tr8 (16-lane group g: channels
    // 16 (g & 1) .., pixels 8 (g >> 1) ..)
    const unsigned char* bx[5];
    unsigned bsh[5];                                                       // odd first element: shift the five dwords by 2 bytes
#pragma unroll
    for (int nb = 0; nb < 5; ++nb) {
        const int n = min(nb * 32 + (lane & 31), 146);
        const int kh_ = n / 21, kw_ = (n % 21) / 3, c_ = n % 3;
        const int o = 8 * khalf + (kw_ >> 1) + 3;
        bx[nb] = xs + ((((size_t)(wave * SW16_ROWS + kh_) * 3 + c_) * 2 + (kw_ & 1)) * SW16_PH + (o & ~1)) * 2;
        bsh[nb] = (o & 1) * 2;
    }
    const int g = lane >> 4;
    const unsigned char* ga = gs + (8 * (g >> 1) + ((lane & 15) >> 2)) * 64 + (g & 1) * 32 + (lane & 3) * 8;

    // a contiguous strip of tiles per workgroup
    const long long per = (p.tiles + gridDim.x - 1) / gridDim.x;
    long long tile = (long long)blockIdx.x * per;
    const long long tend = tile + per < p.tiles ? tile + per : p.tiles;
    if (tile < tend) load_tile(tile);
    for (; tile < tend; ++tile) {
        __syncthreads();                                                  // the previous tile's fragment reads are done
        store_tile();
        __syncthreads();
        if (tile + 1 < tend) load_tile(tile + 1);
        // the two output rows of the tile alternate through two fragment sets: the reads of one are in flight under the
        // matrix instructions of the other
        u16x8 a0, a1, b0[5], b1[5];
        auto frags = [&](int ro, int p0, u16x8& a, u16x8 (&b)[5]) {
            const unsigned char* ap = ga + (size_t)(ro * SW16_PMAX + p0) * 64;
            a = lds_tr8(ap, ap + 256);
            const size_t boff = ((size_t)(2 * ro) * SW16_XROW + p0) * 2;
#pragma unroll
            for (int nb = 0; nb < 5; ++nb) b[nb] = lds_read8_shifted(bx[nb] + boff, bsh[nb]);
        };
        static_assert(SW16_R == 2, "the fragment pipeline below alternates two rows");
        frags(0, 0, a0, b0);
        for (int p0 = 0; p0 < P; p0 += 16) {
            frags(1, p0, a1, b1);
#pragma unroll
            for (int nb = 0; nb < 5; ++nb) mma_k16(a0, b0[nb], acc[nb], T());
            if (p0 + 16 < P) frags(0, p0 + 16, a0, b0);
#pragma unroll
            for (int nb = 0; nb < 5; ++nb) mma_k16(a1, b1[nb], acc[nb], T());
        }
    }
    f32x4* out = (f32x4*)p.ws + ((((size_t)blockIdx.x * gridDim.y + blockIdx.y) * 7 + wave) * 5) * 4 * 64 + lane;
#pragma unroll
    for (int nb = 0; nb < 5; ++nb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = {acc[nb][4 * q], acc[nb][4 * q + 1], acc[nb][4 * q + 2], acc[nb][4 * q + 3]};
            out[(nb * 4 + q) * 64] = v;
        }
}

// sums the partial tiles over the gx workgroups of a channel block (fixed order) and scatters them into dw [Cout][3][7][7][7]
__global__ void stem_wgrad16_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int gx, int gy, int Cout, int accumulate) {
    const long long per_x = (long long)gy * 35 * 4 * 64;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < per_x; idx += (long long)blockDim.x * gridDim.x) {
        long long t = idx;
        const int lane = (int)(t % 64); t /= 64;
        const int q = (int)(t % 4); t /= 4;
        const int nb = (int)(t % 5); t /= 5;
        const int kd = (int)(t % 7);
        const int by = (int)(t / 7);
        const int n = nb * 32 + (lane & 31);
        if (n >= 147) continue;
        const int kh = n / 21, kw = (n % 21) / 3, c = n % 3;
        f32x4 sum = {0.f, 0.f, 0.f, 0.f};
        for (int x = 0; x < gx; ++x) {
            const f32x4 v = ((const f32x4*)ws)[(size_t)x * per_x + idx];
            sum[0] += v[0]; sum[1] += v[1]; sum[2] += v[2]; sum[3] += v[3];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = by * 32 + cd_row(4 * q + r, lane);
            if (co < Cout) {
                float* o = dw + ((((size_t)co * 3 + c) * 7 + kd) * 7 + kh) * 7 + kw;
                *o = accumulate ? *o + sum[r] : sum[r];
            }
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
// With a workspace (partial tiles + fixed-order sum) a job ends in plain stores, not atomics, and parallelism wins again: swept on the
// C4 step with the workspace forms (round 3, ms per step): fp32 MFMA 32 px -> 45.2, 64 -> 43.0, 128 -> 42.2, 256 -> 43.0, 512 -> 46.0,
// 1024 -> 49.6; 16-bit per-tap form 64 -> 19.5, 128 -> 18.40, 256 -> 18.31, 512 -> 18.55.  Once a workgroup sums its four jobs
// before writing (one partial tile per workgroup): fp32 32 px -> 41.1, 64 -> 41.1, 96 -> 41.4, 128 -> 41.7; 16-bit 64...256 all 18.0.
enum { WG_ATOMICS = 0, WG_WS_F32 = 1, WG_WS_16 = 2 };
static int wgrad_min_pixels(int form) {
    const int x = opt(STEP_OPT_WGRAD_MINPIX);                  // (tests exercise both regimes in one process)
    return x > 0 ? (x + 15) / 16 * 16 : (form == WG_WS_F32 ? 64 : (form == WG_WS_16 ? 256 : 512));
}

// launch plan of the LDS-tiled 16-bit form; ok = false: the shape is left to the per-tap forms
struct Wg16Plan { bool ok = false, pw = false, wide = false, nw12 = false; int rows, cpp, upj, cot, cit; long long units, gx, gy; };
static Wg16Plan wgrad16_plan(const step_conv_desc* d) {
    Wg16Plan pl; pl.ok = false; pl.pw = false; pl.wide = false; pl.rows = pl.cpp = pl.upj = pl.cot = pl.cit = 0; pl.units = pl.gx = pl.gy = 0;
    if (!d || (d->dtype != STEP_BF16 && d->dtype != STEP_F16)) return pl;
    const bool pw = d->kd == 1 && d->kh == 1 && d->kw == 1;
    pl.pw = pw;
    if (!pw && !(d->kh == 3 && d->kw == 3 && (d->kd == 1 || d->kd == 3))) return pl;
    if (d->Cin % 8 || d->Cout % 8 || d->x_cstride % 8 || d->x_coff % 8 || d->y_cstride % 8 || d->y_coff % 8) return pl;
    if (opt(STEP_OPT_WGRAD16_LDS) == 0) return pl;             // (1 and 2 differ only for the pointwise layers: wgradpws_plan)
    if (d->N <= 0 || d->D <= 0 || d->H <= 0 || d->W <= 0) return pl;
    if (pw) {
        const long long M = (long long)d->N * d->D * d->H * d->W;
        // measured (tools/wgrad_bench.py, bf16): 480 -> 304 channels on 25x25x9: 56 -> 37 us; 256 -> 288 on 50x50x18: 74 -> 99 us; 64 -> 64
        // on 100x100x18: 53 -> 154 us -- the six waves want six full 32-channel blocks: deep-Cin layers only
#ifndef WG16_PW_MINCIN
#define WG16_PW_MINCIN 384
#endif
        if (M < 4 * WG16_PW_P || d->Cin < WG16_PW_MINCIN) return pl;
        pl.cot = ceil_div(d->Cout, 64); pl.cit = ceil_div(d->Cin, 192);
        pl.cpp = 1; pl.rows = 1;
        pl.units = ceil_div64(M, WG16_PW_P);
        pl.gy = (long long)pl.cot * pl.cit;
    } else {
        int R = 0, Rw = 0;
        for (int r = d->H; r >= 1; --r) {
            if (!R && r * d->W <= WG16_P && (r + 2) * (d->W + 2) <= WG16_XP) R = r;
            if (!Rw && r * d->W <= WG16_P && (r + 2) * (d->W + 2) <= WG16_XP_WIDE) Rw = r;
        }
        // the wide-halo instantiation only where it cuts the chunks per plane (measured, bf16, 8 clips: conv3d_2c on 100x100 2.61 -> 1.80 ms,
        // on 56x56 0.53 -> 0.48 ms; the 50- and 25-wide layers keep the same rows and would lose 5 % to its longer staging loops)
        if (Rw > 0 && (R <= 0 || ceil_div(d->H, Rw) < ceil_div(d->H, R))) { R = Rw; pl.wide = true; }
#ifndef WG16_NW6
        pl.nw12 = true;
#endif
        if (R <= 0) return pl;
        pl.cot = ceil_div(d->Cout, 64); pl.cit = ceil_div(d->Cin, 64);
        pl.cpp = ceil_div(d->H, R);
        pl.rows = ceil_div(d->H, pl.cpp);                    // balanced chunks
        pl.units = (long long)d->N * d->D * pl.cpp;
        pl.gy = (long long)d->kd * pl.cot * pl.cit;
    }
    // How many workgroups along the pixel axis (each walks `upj` (plane, chunk) units, the next one's loads under this one's matrix
    // work).  The 3x3 form holds ONE workgroup per CU, the pointwise form two: a grid just over a whole number of rounds leaves most
    // of the chip idle for a whole workgroup time (measured with fixed targets of 256 / 512 / 1024 workgroups, 8 AVA clips:
    // 3c_b1b 1.57 / 1.24 / 0.90 ms, 4f_b1b 0.35 / 0.46 / 0.44 ms, 5c on 1080 7x7 maps 0.99 / 0.89 / 0.60 ms -- the winner is whichever
    // lands just under a multiple of 256).  So the count is chosen per layer: time ~ rounds x units per workgroup, plus the partial
    // tiles every pixel-axis workgroup writes and the sum reads back (~150 KB each way per workgroup: ~1 % of a unit's time each).
#ifndef WG16_WGCOST
#define WG16_WGCOST 0.012       /* a workgroup's partial tiles (written here, read back by the sum), in units of a unit's time */
#endif
#ifdef WG16_TOTAL           /* experiment builds: a fixed target, as rounds 2-3 had (512) */
    {
        long long want = WG16_TOTAL / (pl.gy > 0 ? pl.gy : 1);
        if (want < 1) want = 1;
        long long upj = ceil_div64(pl.units, want);
        if (upj < 1) upj = 1;
        pl.upj = (int)(upj > 0x3fffffff ? 0x3fffffff : upj);
        pl.gx = ceil_div64(pl.units, pl.upj);
        if (pl.gx >= 8) pl.gx = (pl.gx + 7) / 8 * 8;
    }
#else
    {
        const long long slots = 256LL * (pw ? 2 : 1);
        double best = -1.0;
        long long best_gx = 1, best_upj = pl.units;
        const long long gx_max = pl.units < 2048 / (pl.gy > 0 ? pl.gy : 1) + 8 ? pl.units : 2048 / (pl.gy > 0 ? pl.gy : 1) + 8;
        for (long long cand = 1; cand <= (gx_max > 1 ? gx_max : 1); cand = cand < 8 ? cand + 1 : cand + 8) {
            const long long upj = ceil_div64(pl.units, cand);
            long long gx = ceil_div64(pl.units, upj);
            if (gx >= 8) gx = (gx + 7) / 8 * 8;              // whole rounds over the 8 XCDs (the kernel's launch-order remap); surplus workgroups write zero tiles
            const long long wgs = gx * pl.gy;
            const double cost = (double)ceil_div64(wgs, slots) * (double)upj + WG16_WGCOST * (double)wgs / (double)slots * 256.0 + 0.25;   // (+ a launch's fixed part)
            if (best < 0.0 || cost < best) { best = cost; best_gx = gx; best_upj = upj; }
        }
        pl.upj = (int)(best_upj > 0x3fffffff ? 0x3fffffff : best_upj);
        pl.gx = best_gx;
    }
#endif
    pl.ok = pl.gy <= 65535 && pl.gx <= 0x7fffffffLL;
    return pl;
}

// job split of the per-tap kernel (conv_wgrad_kernel): pointwise layers cut the flattened pixel axis into `chunk`-pixel jobs (+ one
// ragged tail job), windows cut the (n, d, h) rows; gy = (tap, co tile, ci tile) workgroup rows; per_tile = floats of a job's tile
struct WgJobs { bool pw; int chunk, tail, rows; long long full, jobs, gy; int cot, cit, nbw; size_t per_tile; };
static WgJobs wgrad_jobs(const step_conv_desc* d, int form) {
    WgJobs j;
    const int ntaps = d->kd * d->kh * d->kw;
    const bool narrow = d->Cin <= 32;
    j.nbw = narrow ? 1 : 2;
    j.cot = ceil_div(d->Cout, 64); j.cit = ceil_div(d->Cin, narrow ? 32 : 64);
    j.per_tile = (size_t)2 * j.nbw * 16 * 64;
    j.pw = ntaps == 1;
    j.chunk = j.tail = j.rows = 0; j.full = 0;
    if (j.pw) {
        // pointwise: no neighbourhood, so the pixel axis is cut into chunks ("rows" of one long plane list); ~6000 wavefront jobs per
        // launch (see below), a multiple of the 16-pixel MFMA step
        const long long M = (long long)d->N * d->D * d->H * d->W;
        constexpr int wg_jobs_pw = 6144;
        const long long tiles = (long long)j.cot * j.cit;
        long long want = wg_jobs_pw / (tiles > 0 ? tiles : 1);
        if (want < 1) want = 1;
        long long ch = (ceil_div64(M, want) + 15) / 16 * 16;
        if (ch < wgrad_min_pixels(form)) ch = wgrad_min_pixels(form);
        if (ch > 65536) ch = 65536;
        j.chunk = (int)ch;
        j.full = M / j.chunk;                                   // (n, d, h) collapse into full chunks; the ragged tail is a second launch
        j.tail = (int)(M % j.chunk);
        j.jobs = j.full + (j.tail ? 1 : 0);
        j.gy = tiles;
        return j;
    }
    j.gy = (long long)ntaps * j.cot * j.cit;
    // (n, d, h) rows per wavefront job.  Two opposite pressures (PMC): the kernel hides its load latency only with
    // several wavefronts per SIMD (1.6 per SIMD -> matrix pipe 18 % busy), but every job ends in one pass over its 64x64 tile
    // (4096 fp32 atomics, or 16 KiB of partial tile).  Aim at ~6000 wavefront jobs per launch, whatever the map size.
    constexpr int wg_jobs = 6144;
    const long long total_rows = (long long)d->N * d->D * d->H;
    long long want = wg_jobs / (j.gy > 0 ? j.gy : 1);
    if (want < 1) want = 1;
    long long rows = ceil_div64(total_rows, want);
    const long long rows_min = ceil_div64(wgrad_min_pixels(form), d->W);      // small maps: fewer, longer jobs (see above)
    if (rows < rows_min) rows = rows_min;
    if (rows > total_rows) rows = total_rows;
    if (rows < 1) rows = 1;
    if (rows > 0x3fffffff) rows = 0x3fffffff;
    j.rows = (int)rows;
    j.jobs = ceil_div64(total_rows, j.rows);
    return j;
}
// one partial tile per WORKGROUP (four jobs): pointwise layers launch their full chunks and the ragged tail separately
static long long wgrad_ws_blocks(const WgJobs& j) { return j.pw ? ceil_div64(j.full, 4) + (j.tail ? 1 : 0) : ceil_div64(j.jobs, 4); }
static size_t wgrad_jobs_ws_bytes(const WgJobs& j) { return (size_t)wgrad_ws_blocks(j) * (size_t)j.gy * j.per_tile * sizeof(float); }

// defer != NULL: the fixed-order sum of the partial tiles is NOT launched; *defer describes it (step_wgrad_reduce_group runs several
// layers' sums as one launch).  kind 0: nothing is pending (the atomics forms, an empty batch).

// the pixel-stream form of the pointwise weight gradient (conv_wgrad16_pws_kernel): tiles, slices
struct WgPwsPlan { bool ok = false; int cot = 0, cit = 0, co_t = 0, ci_t = 0; long long gx = 0, gy = 0, ppj = 0; };
static WgPwsPlan wgradpws_plan(const step_conv_desc* d) {
    WgPwsPlan pl;
    if (!d || (d->dtype != STEP_BF16 && d->dtype != STEP_F16)) return pl;
    if (!(d->kd == 1 && d->kh == 1 && d->kw == 1)) return pl;
    if (d->Cin % 8 || d->Cout % 8 || d->x_cstride % 8 || d->x_coff % 8 || d->y_cstride % 8 || d->y_coff % 8) return pl;
    if (opt(STEP_OPT_WGRAD16_LDS) != 1) return pl;             // 2: the pointwise layers on the forms of the start of round 4 (A/B, tests); 0: per-tap everywhere
    if (d->N <= 0 || d->D <= 0 || d->H <= 0 || d->W <= 0) return pl;
    const long long M = (long long)d->N * d->D * d->H * d->W;
#ifndef WGPWS_MINPIX
#define WGPWS_MINPIX 2048
#endif
    if (M < WGPWS_MINPIX) return pl;
    pl.cot = ceil_div(d->Cout, 128); pl.co_t = (ceil_div(d->Cout, pl.cot) + 31) / 32 * 32;
    pl.cit = ceil_div(d->Cin, 256); pl.ci_t = (ceil_div(d->Cin, pl.cit) + 63) / 64 * 64;
    pl.gy = (long long)pl.cot * pl.cit;
    const long long stages = ceil_div64(M, PWS_P);
    // slices: time ~ rounds of the chip (two workgroups per CU) x (stages per slice + the slice's image, written here and read back by the
    // sum, in units of a stage's operand bytes)
    const double img = 2.0 * pl.co_t * pl.ci_t * 4.0 / (double)(PWS_P * (pl.co_t + pl.ci_t) * 2);
    const long long slots = 512;
    double best = -1.0;
    long long best_gx = 1;
    const long long gx_max = stages / 4 > 1 ? stages / 4 : 1;
    for (long long cand = 1; cand <= gx_max && cand * pl.gy <= 4096; cand = cand < 8 ? cand + 1 : cand + 8) {
        const long long spj = ceil_div64(stages, cand);
        long long gx = ceil_div64(stages, spj);
        if (gx >= 8) gx = (gx + 7) / 8 * 8;
        const double cost = (double)ceil_div64(gx * pl.gy, slots) * ((double)spj + img) + 2.0;
        if (best < 0.0 || cost < best) { best = cost; best_gx = gx; }
    }
    pl.gx = best_gx;
    pl.ppj = ceil_div64(stages, pl.gx) * PWS_P;               // whole stages per slice, every pixel covered (surplus slices write zero images)
    // the kernel addresses a slice's operands with 32-bit element offsets from the slice's first pixel
    const long long span = (pl.ppj + PWS_P) * (long long)(d->x_cstride > d->y_cstride ? d->x_cstride : d->y_cstride);
    pl.ok = pl.gy <= 65535 && pl.gx <= 0x7fffffffLL && pl.gx > 0 && span < 0x7fffffffLL;
    return pl;
}

static void reduce_item_none(step_wgrad_reduce_item* it) { if (it) { *it = step_wgrad_reduce_item(); } }

static int conv_wgrad_impl(const step_conv_desc* d, const void* x, const void* dy, bool w16, float* dw, int accumulate, void* ws,
                           size_t ws_bytes, step_stream_t stream, step_wgrad_reduce_item* defer = nullptr) {
    reduce_item_none(defer);
    if (!d) return STEP_E_NULL;
    if (w16 && d->dtype != STEP_BF16 && d->dtype != STEP_F16) return STEP_E_UNSUPPORTED;
    if (d->N < 0 || d->D <= 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->Cout <= 0) return STEP_E_SHAPE;
    if (d->kd <= 0 || d->kh <= 0 || d->kw <= 0 || !(d->kd & 1) || !(d->kh & 1) || !(d->kw & 1)) return STEP_E_UNSUPPORTED;
    if (d->x_coff < 0 || d->x_coff + d->Cin > d->x_cstride || d->y_coff < 0 || d->y_coff + d->Cout > d->y_cstride) return STEP_E_SHAPE;
    if (!dw) return STEP_E_NULL;
    const int ntaps = d->kd * d->kh * d->kw;
    const bool pws_form = w16 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)dy % 16) == 0 && d->N > 0 && wgradpws_plan(d).ok;
    const bool lds_form = pws_form || (w16 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)dy % 16) == 0 && wgrad16_plan(d).ok && d->N > 0);
    // the per-tap kernel with a workspace: partial tiles + a fixed-order sum write every element of dw themselves; its jobs are
    // shorter than the atomics form's (wgrad_min_pixels)
    const WgJobs jws = wgrad_jobs(d, w16 ? WG_WS_16 : WG_WS_F32);
    const bool tap_ws = !lds_form && d->N > 0 && ws && ((uintptr_t)ws % 16) == 0 && ws_bytes >= wgrad_jobs_ws_bytes(jws) && jws.jobs > 0;
    const WgJobs jb = tap_ws ? jws : wgrad_jobs(d, WG_ATOMICS);
    if (!accumulate && !tap_ws) {
        const int e = (int)hipMemsetAsync(dw, 0, (size_t)d->Cout * d->Cin * ntaps * sizeof(float), (hipStream_t)stream);
        if (e != 0) return e;
    }
    if (d->N == 0) return STEP_OK;
    if (!x || !dy) return STEP_E_NULL;
    WgradParams p;
    p.x = x; p.dy = dy; p.dw = dw; p.ws = tap_ws ? (float*)ws : nullptr;
    p.N = d->N; p.D = d->D; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Cout = d->Cout; p.kd = d->kd; p.kh = d->kh; p.kw = d->kw;
    p.x_cstride = d->x_cstride; p.x_coff = d->x_coff; p.dy_cstride = d->y_cstride; p.dy_coff = d->y_coff;
    // pointwise layers: the pixel-stream form (wgradpws_plan)
    if (pws_form) {
        const WgPwsPlan pp = wgradpws_plan(d);
        WgradPwsParams q;
        q.x = x; q.dy = dy; q.dw = dw;
        q.M = (long long)d->N * d->D * d->H * d->W; q.ppj = pp.ppj;
        q.Cin = d->Cin; q.Cout = d->Cout; q.x_cstride = d->x_cstride; q.x_coff = d->x_coff; q.dy_cstride = d->y_cstride; q.dy_coff = d->y_coff;
        q.cot = pp.cot; q.cit = pp.cit; q.co_t = pp.co_t; q.ci_t = pp.ci_t;
        const size_t need = (size_t)pp.gx * d->Cout * d->Cin * sizeof(float);
        q.ws = (ws && ws_bytes >= need && ((uintptr_t)ws % 16) == 0 && ((uintptr_t)dw % 16) == 0) ? (float*)ws : nullptr;
        const dim3 gridp((unsigned)pp.gx, (unsigned)pp.gy);
        if (d->dtype == STEP_BF16) STEP_LAUNCH((conv_wgrad16_pws_kernel<bf16_t>), gridp, dim3(512), stream, q);
        else STEP_LAUNCH((conv_wgrad16_pws_kernel<f16_t>), gridp, dim3(512), stream, q);
        if (q.ws && defer) {
            defer->kind = 3; defer->ws = q.ws; defer->dw = dw; defer->jobs = pp.gx; defer->Cout = d->Cout; defer->Cin = d->Cin; defer->taps = 1; defer->accumulate = 1;
        } else if (q.ws) {
            const long long n4 = (long long)d->Cout * d->Cin / 4;
            STEP_LAUNCH(wgradpws_reduce_kernel, dim3(flat_grid((n4 + 15) / 16 * 256, 256)), dim3(256), stream, (const float*)q.ws, dw, pp.gx, n4, 1);   // (dw was cleared above unless accumulate)
        }
        return STEP_LAUNCH_CHECK();
    }
    // the LDS-tiled form (transpose reads): 3x3 windows or pointwise, channel counts in 16-byte vectors (wgrad16_plan)
    if (w16 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)dy % 16) == 0) {
        const Wg16Plan pl = wgrad16_plan(d);
        if (pl.ok) {
            Wgrad16Params q;
            q.x = x; q.dy = dy; q.dw = dw;
            q.N = d->N; q.D = d->D; q.H = d->H; q.W = d->W; q.Cin = d->Cin; q.Cout = d->Cout; q.kd = d->kd;
            q.x_cstride = d->x_cstride; q.x_coff = d->x_coff; q.dy_cstride = d->y_cstride; q.dy_coff = d->y_coff;
            q.cot = pl.cot; q.cit = pl.cit; q.cpp = pl.cpp; q.rows = pl.rows; q.units = pl.units; q.upj = pl.upj;
            q.wmagic = (unsigned)(((1u << 22) + d->W - 1) / d->W);
            q.wmagic2 = (unsigned)(((1u << 22) + d->W + 1) / (d->W + 2));
            const size_t need = (size_t)pl.gx * pl.gy * 6 * (pl.pw ? 8 : 24) * 64 * 16;
            q.ws = (ws && ws_bytes >= need && ((uintptr_t)ws % 16) == 0) ? (float*)ws : nullptr;
            dim3 grid16((unsigned)pl.gx, (unsigned)pl.gy);
#ifdef WG16_NOPF            /* experiment builds (make EXP=... EXPFLAGS=-DWG16_NOPF): the two-workgroup form for the 3x3 windows too */
#define WG16_K3 conv_wgrad16_lds2_kernel
#else
#define WG16_K3 conv_wgrad16_lds_kernel
#endif
            if (pl.pw) {
                if (d->dtype == STEP_BF16) STEP_LAUNCH((conv_wgrad16_lds2_kernel<bf16_t, 1>), grid16, dim3(384), stream, q);
                else STEP_LAUNCH((conv_wgrad16_lds2_kernel<f16_t, 1>), grid16, dim3(384), stream, q);
            } else {
                if (pl.wide && pl.nw12) {
                    if (d->dtype == STEP_BF16) STEP_LAUNCH((conv_wgrad16_lds12_wide_kernel<bf16_t, 3>), grid16, dim3(768), stream, q);
                    else STEP_LAUNCH((conv_wgrad16_lds12_wide_kernel<f16_t, 3>), grid16, dim3(768), stream, q);
                } else if (pl.wide) {
                    if (d->dtype == STEP_BF16) STEP_LAUNCH((conv_wgrad16_lds_wide_kernel<bf16_t, 3>), grid16, dim3(384), stream, q);
                    else STEP_LAUNCH((conv_wgrad16_lds_wide_kernel<f16_t, 3>), grid16, dim3(384), stream, q);
                } else if (pl.nw12) {
                    if (d->dtype == STEP_BF16) STEP_LAUNCH((conv_wgrad16_lds12_kernel<bf16_t, 3>), grid16, dim3(768), stream, q);
                    else STEP_LAUNCH((conv_wgrad16_lds12_kernel<f16_t, 3>), grid16, dim3(768), stream, q);
                } else if (d->dtype == STEP_BF16) STEP_LAUNCH((WG16_K3<bf16_t, 3>), grid16, dim3(384), stream, q);
                else STEP_LAUNCH((WG16_K3<f16_t, 3>), grid16, dim3(384), stream, q);
            }
#undef WG16_K3
            if (q.ws && defer) {
                defer->kind = 2; defer->ws = q.ws; defer->dw = dw; defer->jobs = pl.gx; defer->gy = (int)pl.gy; defer->cot = pl.cot; defer->cit = pl.cit;
                defer->Cout = d->Cout; defer->Cin = d->Cin; defer->taps = d->kd; defer->accumulate = 1; defer->pw = pl.nw12 ? 2 : (int)pl.pw;
            } else if (q.ws) {
                const long long groups = pl.gy * 6 * (pl.pw ? 8 : 24) * 64;
                STEP_LAUNCH(wgrad16_reduce_kernel, dim3(flat_grid(groups * 4, 256)), dim3(256), stream, (const float*)q.ws, dw, (int)pl.gx, (int)pl.gy,
                            pl.cot, pl.cit, d->Cout, d->Cin, d->kd, 1, pl.nw12 ? 2 : (int)pl.pw);      // (dw was cleared above unless accumulate)
            }
            return STEP_LAUNCH_CHECK();
        }
    }
    if (ntaps == 1) {
        const long long M = (long long)d->N * d->D * d->H * d->W;
        if (M > 0x7fffffffLL) return STEP_E_UNSUPPORTED;
        const int chunk = jb.chunk;
        const long long full = jb.full;
        const int tail = jb.tail;
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
        if (rc == STEP_OK && tail) {
            if (tap_ws) p.ws = (float*)ws + (size_t)ceil_div64(full, 4) * jb.gy * jb.per_tile;      // the tail job's tiles behind the full chunks
            launch(1, tail, (size_t)full * chunk);
        }
        if (rc == STEP_OK && tap_ws && defer) {
            defer->kind = 1; defer->ws = (const float*)ws; defer->dw = dw; defer->jobs = wgrad_ws_blocks(jb); defer->gy = (int)jb.gy; defer->nbw = jb.nbw;
            defer->cot = jb.cot; defer->cit = jb.cit; defer->Cout = d->Cout; defer->Cin = d->Cin; defer->taps = 1; defer->accumulate = accumulate; defer->pw = 1;
        } else if (rc == STEP_OK && tap_ws)
            STEP_LAUNCH(wgrad_reduce_kernel, dim3(flat_grid(jb.gy * (long long)jb.per_tile * 4, 256)), dim3(256), stream, (const float*)ws, dw, wgrad_ws_blocks(jb),
                        (int)jb.gy, 2, jb.nbw, jb.cot, jb.cit, d->Cout, d->Cin, 1, accumulate);
        return rc != STEP_OK ? rc : STEP_LAUNCH_CHECK();
    }
    const bool narrow = d->Cin <= 32;
    p.cot = jb.cot; p.cit = jb.cit;
    // (a tap-row form -- a job owns a row of three kw taps and re-pairs ten loaded pixels in registers -- measured 1.3-2.6x SLOWER
    // than this per-tap form: 424 VGPRs, one wave per SIMD; removed)
    const long long gy = jb.gy;
    p.total_rows = (long long)d->N * d->D * d->H;
    p.rows = jb.rows;
    p.jobs = jb.jobs;
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
    if (tap_ws && defer) {
        defer->kind = 1; defer->ws = (const float*)ws; defer->dw = dw; defer->jobs = wgrad_ws_blocks(jb); defer->gy = (int)jb.gy; defer->nbw = jb.nbw;
        defer->cot = jb.cot; defer->cit = jb.cit; defer->Cout = d->Cout; defer->Cin = d->Cin; defer->taps = ntaps; defer->accumulate = accumulate; defer->pw = 0;
    } else if (tap_ws)
        STEP_LAUNCH(wgrad_reduce_kernel, dim3(flat_grid(jb.gy * (long long)jb.per_tile * 4, 256)), dim3(256), stream, (const float*)ws, dw, wgrad_ws_blocks(jb),
                    (int)jb.gy, 2, jb.nbw, jb.cot, jb.cit, d->Cout, d->Cin, ntaps, accumulate);
    return STEP_LAUNCH_CHECK();
}

size_t step_conv_wgrad_workspace_bytes(const step_conv_desc* d) {
    if (!d || d->N <= 0 || d->D <= 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->kd <= 0 || d->kh <= 0 || d->kw <= 0) return 0;
    return wgrad_jobs_ws_bytes(wgrad_jobs(d, WG_WS_F32));
}

int step_conv_wgrad_ws(const step_conv_desc* d, const void* x, const float* dy, float* dw, int accumulate, void* ws, size_t ws_bytes,
                       step_stream_t stream) {
    // a scratch that is given must be usable: a short or misaligned one is an error, not a silent return to atomics
    if (ws && d && (ws_bytes < step_conv_wgrad_workspace_bytes(d) || ((uintptr_t)ws % 16) != 0)) return STEP_E_SHAPE;
    return conv_wgrad_impl(d, x, dy, false, dw, accumulate, ws, ws_bytes, stream);
}

int step_conv_wgrad(const step_conv_desc* d, const void* x, const float* dy, float* dw, int accumulate, step_stream_t stream) {
    return conv_wgrad_impl(d, x, dy, false, dw, accumulate, nullptr, 0, stream);
}

int step_conv_wgrad16(const step_conv_desc* d, const void* x, const void* dy, float* dw, int accumulate, step_stream_t stream) {
    return conv_wgrad_impl(d, x, dy, true, dw, accumulate, nullptr, 0, stream);
}

size_t step_conv_wgrad16_workspace_bytes(const step_conv_desc* d) {
    if (!d || (d->dtype != STEP_BF16 && d->dtype != STEP_F16)) return 0;
    const WgPwsPlan pp = wgradpws_plan(d);
    if (pp.ok) return (size_t)pp.gx * d->Cout * d->Cin * sizeof(float);
    const Wg16Plan pl = wgrad16_plan(d);
    if (pl.ok) return (size_t)pl.gx * pl.gy * 6 * (pl.pw ? 8 : 24) * 64 * 16;
    if (d->N <= 0 || d->D <= 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->kd <= 0 || d->kh <= 0 || d->kw <= 0) return 0;
    return wgrad_jobs_ws_bytes(wgrad_jobs(d, WG_WS_16));         // the per-tap 16-bit form: partial tiles of its wavefront jobs
}

int step_conv_wgrad16_ws(const step_conv_desc* d, const void* x, const void* dy, float* dw, int accumulate, void* ws, size_t ws_bytes,
                         step_stream_t stream) {
    return conv_wgrad_impl(d, x, dy, true, dw, accumulate, ws, ws_bytes, stream);
}


int step_conv_wgrad_kernel_name(const step_conv_desc* d, int dy16, char* buf, int buflen) {
    if (!d || !buf || buflen <= 0) return STEP_E_NULL;
    const char* t = d->dtype == STEP_F32 ? "float" : (d->dtype == STEP_BF16 ? "step::bf16_t" : "step::f16_t");
    const Wg16Plan pl = dy16 ? wgrad16_plan(d) : Wg16Plan();
    if (dy16 && wgradpws_plan(d).ok) snprintf(buf, (size_t)buflen, "void step::conv_wgrad16_pws_kernel<%s>(step::WgradPwsParams)", t);
    else if (dy16 && pl.ok) {
        const bool pw = d->kd == 1 && d->kh == 1 && d->kw == 1;
        snprintf(buf, (size_t)buflen, "void step::%s<%s, %d>(step::Wgrad16Params)",
                 pw ? "conv_wgrad16_lds2_kernel" : (pl.nw12 ? (pl.wide ? "conv_wgrad16_lds12_wide_kernel" : "conv_wgrad16_lds12_kernel")
                                                             : (pl.wide ? "conv_wgrad16_lds_wide_kernel" : "conv_wgrad16_lds_kernel")), t, pw ? 1 : 3);
    } else
        snprintf(buf, (size_t)buflen, "void step::conv_wgrad_kernel<%s, 2, %d, %s>(step::WgradParams)", t, d->Cin <= 32 ? 1 : 2, dy16 ? "true" : "false");
    return STEP_OK;
}

int step_conv_wgrad_partial(const step_conv_desc* d, const void* x, const void* dy, int dy16, float* dw, int accumulate, void* ws, size_t ws_bytes,
                            step_wgrad_reduce_item* item, step_stream_t stream) {
    if (!item) return STEP_E_NULL;
    if (ws && d && ((uintptr_t)ws % 16) != 0) return STEP_E_SHAPE;
    if (ws && d && ws_bytes < (dy16 ? step_conv_wgrad16_workspace_bytes(d) : step_conv_wgrad_workspace_bytes(d))) return STEP_E_SHAPE;
    return conv_wgrad_impl(d, x, dy, dy16 != 0, dw, accumulate, ws, ws_bytes, stream, item);
}

static int wgrad_reduce_group_launch(const WgradReduceGroup& g, int m, long long most, step_stream_t stream) {
    STEP_LAUNCH(wgrad_reduce_group_kernel, dim3(flat_grid(most, 256), (unsigned)m), dim3(256), stream, g);
    return STEP_LAUNCH_CHECK();
}

int step_wgrad_reduce_group(const step_wgrad_reduce_item* items, int n, step_stream_t stream) {
    if (n < 0 || n > STEP_WGRAD_REDUCE_MAX) return STEP_E_SHAPE;
    if (n == 0) return STEP_OK;
    if (!items) return STEP_E_NULL;
    for (int i = 0; i < n; ++i) {
        const step_wgrad_reduce_item& it = items[i];
        if (it.kind == 0) continue;
        if ((it.kind != 1 && it.kind != 2 && it.kind != 3) || !it.ws || !it.dw) return STEP_E_SHAPE;
    }
    // The members of one launch run CONCURRENTLY (blockIdx.y) and each ends in a plain read-modify-write of its dw: two sums into the
    // same gradient (a unit used twice in one graph, leftovers of an earlier backward) must not share a launch.  A repeated dw closes the
    // launch; the stream then orders the two sums as separate launches did before the grouping.
    WgradReduceGroup g;
    int m = 0;
    long long most = 0;
    for (int i = 0; i < n; ++i) {
        const step_wgrad_reduce_item& it = items[i];
        if (it.kind == 0) continue;
        bool dup = false;
        for (int q = 0; q < m; ++q) dup = dup || g.it[q].dw == it.dw;
        if (dup) {
            for (int q = m; q < STEP_WGRAD_REDUCE_MAX; ++q) g.it[q] = step_wgrad_reduce_item();
            const int rc = wgrad_reduce_group_launch(g, m, most, stream);
            if (rc != STEP_OK) return rc;
            m = 0; most = 0;
        }
        const long long work = it.kind == 1 ? (long long)it.gy * (2 * it.nbw * 16 * 64) * 4
                             : (it.kind == 2 ? (long long)it.gy * 6 * (it.pw == 1 ? 8 : 24) * 64 * 4 : ((long long)it.Cout * it.Cin / 4 + 15) / 16 * 256);
        if (work > most) most = work;
        g.it[m++] = it;
    }
    if (m == 0) return STEP_OK;
    for (int i = m; i < STEP_WGRAD_REDUCE_MAX; ++i) g.it[i] = step_wgrad_reduce_item();
    return wgrad_reduce_group_launch(g, m, most, stream);
}


struct StemJobs { int To, Ho, Wo, cot, rows, hchunks; long long jobs; };
static StemJobs stem_wgrad_jobs(int N, int T, int H, int W, int Cout) {
    StemJobs j;
    j.To = (T + 5 - 7) / 2 + 1; j.Ho = (H + 5 - 7) / 2 + 1; j.Wo = (W + 5 - 7) / 2 + 1;
    j.cot = ceil_div(Cout, 64);
    j.rows = 8; j.hchunks = ceil_div(j.Ho > 0 ? j.Ho : 1, j.rows);       // 8 output rows per wavefront job: enough jobs for one clip
    j.jobs = (long long)N * (j.To > 0 ? j.To : 0) * j.hchunks;
    return j;
}

size_t step_stem_wgrad_workspace_bytes(int N, int T, int H, int W, int Cout) {
    if (N <= 0 || T <= 0 || H <= 0 || W <= 0 || Cout <= 0) return 0;
    const StemJobs j = stem_wgrad_jobs(N, T, H, W, Cout);
    if (j.To <= 0 || j.Ho <= 0 || j.Wo <= 0) return 0;
    return (size_t)j.jobs * 49 * j.cot * 32 * 64 * sizeof(float);
}

static int stem_wgrad_impl(int dtype, const void* x, int N, int T, int H, int W, const float* dy, int Cout, float* dw, int accumulate,
                           void* ws, size_t ws_bytes, step_stream_t stream) {
    if (N < 0 || T <= 0 || H <= 0 || W <= 0 || Cout <= 0) return STEP_E_SHAPE;
    if (!dw) return STEP_E_NULL;
    if (ws && N > 0 && (ws_bytes < step_stem_wgrad_workspace_bytes(N, T, H, W, Cout) || ((uintptr_t)ws % 16) != 0)) return STEP_E_SHAPE;
    if (!accumulate && !(ws && N > 0)) {
        const int e = (int)hipMemsetAsync(dw, 0, (size_t)Cout * 3 * 343 * sizeof(float), (hipStream_t)stream);
        if (e != 0) return e;
    }
    if (N == 0) return STEP_OK;
    if (!x || !dy) return STEP_E_NULL;
    StemWgradParams p;
    const StemJobs sj = stem_wgrad_jobs(N, T, H, W, Cout);
    p.x = x; p.dy = dy; p.dw = dw; p.ws = (float*)ws; p.N = N; p.T = T; p.H = H; p.W = W;
    p.To = sj.To; p.Ho = sj.Ho; p.Wo = sj.Wo;
    if (p.To <= 0 || p.Ho <= 0 || p.Wo <= 0) return STEP_E_SHAPE;
    p.Cout = Cout; p.cot = sj.cot;
    p.rows = sj.rows; p.hchunks = sj.hchunks;
    p.jobs = sj.jobs;
    dim3 grid((unsigned)ceil_div64(p.jobs, 4), (unsigned)(49 * p.cot));
    switch (dtype) {
        case STEP_F32: STEP_LAUNCH((stem_wgrad_kernel<float>), grid, dim3(256), stream, p); break;
        case STEP_BF16: STEP_LAUNCH((stem_wgrad_kernel<bf16_t>), grid, dim3(256), stream, p); break;
        case STEP_F16: STEP_LAUNCH((stem_wgrad_kernel<f16_t>), grid, dim3(256), stream, p); break;
        default: return STEP_E_DTYPE;
    }
    if (ws)
        STEP_LAUNCH(stem_wgrad_reduce_kernel, dim3(flat_grid((long long)49 * p.cot * 32 * 64, 256)), dim3(256), stream, (const float*)ws, dw,
                    p.jobs, 49 * p.cot, p.cot, Cout, accumulate);
    return STEP_LAUNCH_CHECK();
}

int step_stem_wgrad(int dtype, const void* x, int N, int T, int H, int W, const float* dy, int Cout, float* dw, int accumulate,
                    step_stream_t stream) {
    return stem_wgrad_impl(dtype, x, N, T, H, W, dy, Cout, dw, accumulate, nullptr, 0, stream);
}

int step_stem_wgrad_ws(int dtype, const void* x, int N, int T, int H, int W, const float* dy, int Cout, float* dw, int accumulate,
                       void* ws, size_t ws_bytes, step_stream_t stream) {
    return stem_wgrad_impl(dtype, x, N, T, H, W, dy, Cout, dw, accumulate, ws, ws_bytes, stream);
}



// launch plan of the 16-bit stem form: ok = false -> the caller keeps step_stem_wgrad
struct Sw16Plan { bool ok; int P, tiles_h, tiles_w, gx, gy; long long tiles; };
static Sw16Plan stem_wgrad16_plan(int dtype, int N, int T, int H, int W, int Cout) {
    Sw16Plan pl; pl.ok = false; pl.P = pl.tiles_h = pl.tiles_w = pl.gx = pl.gy = 0; pl.tiles = 0;
    if (dtype != STEP_BF16 && dtype != STEP_F16) return pl;
    if (N <= 0 || T <= 0 || H <= 0 || W <= 0 || Cout <= 0 || (W % 8) || (Cout % 8)) return pl;      // 16-byte rows of the clip / the gradient
    const int To = (T + 5 - 7) / 2 + 1, Ho = (H + 5 - 7) / 2 + 1, Wo = (W + 5 - 7) / 2 + 1;
    if (To <= 0 || Ho <= 0 || Wo <= 0) return pl;
    pl.tiles_w = ceil_div(Wo, SW16_PMAX);
    pl.P = ceil_div(ceil_div(Wo, pl.tiles_w), 16) * 16;
    pl.tiles_h = ceil_div(Ho, SW16_R);
    pl.tiles = (long long)N * To * pl.tiles_h * pl.tiles_w;
    pl.gy = ceil_div(Cout, 32);
    long long gx = 256 / pl.gy;                                   // one workgroup per CU
    if (gx < 1) gx = 1;
    if (gx > pl.tiles) gx = pl.tiles;
    pl.gx = (int)gx;
    pl.ok = true;
    return pl;
}

size_t step_stem_wgrad16_workspace_bytes(int dtype, int N, int T, int H, int W, int Cout) {
    const Sw16Plan pl = stem_wgrad16_plan(dtype, N, T, H, W, Cout);
    return pl.ok ? (size_t)pl.gx * pl.gy * 35 * 4 * 64 * sizeof(f32x4) : 0;
}

int step_stem_wgrad16(int dtype, const void* x, int N, int T, int H, int W, const void* dy, int Cout, float* dw, int accumulate,
                      void* ws, size_t ws_bytes, step_stream_t stream) {
    if (N < 0 || T <= 0 || H <= 0 || W <= 0 || Cout <= 0) return STEP_E_SHAPE;
    if (dtype != STEP_BF16 && dtype != STEP_F16) return STEP_E_DTYPE;
    if (!dw) return STEP_E_NULL;
    if (N == 0) {
        if (!accumulate) return (int)hipMemsetAsync(dw, 0, (size_t)Cout * 3 * 343 * sizeof(float), (hipStream_t)stream);
        return STEP_OK;
    }
    const Sw16Plan pl = stem_wgrad16_plan(dtype, N, T, H, W, Cout);
    if (!pl.ok) return STEP_E_UNSUPPORTED;
    if (!x || !dy || !ws) return STEP_E_NULL;
    if (ws_bytes < step_stem_wgrad16_workspace_bytes(dtype, N, T, H, W, Cout) || (((uintptr_t)ws) & 15) || (((uintptr_t)x) & 15) ||
        (((uintptr_t)dy) & 15))
        return STEP_E_ALIGN;
    StemWgrad16Params p;
    p.x = x; p.dy = dy; p.ws = (float*)ws; p.N = N; p.T = T; p.H = H; p.W = W;
    p.To = (T + 5 - 7) / 2 + 1; p.Ho = (H + 5 - 7) / 2 + 1; p.Wo = (W + 5 - 7) / 2 + 1;
    p.Cout = Cout; p.P = pl.P; p.tiles_h = pl.tiles_h; p.tiles_w = pl.tiles_w; p.tiles = pl.tiles;
    const dim3 grid((unsigned)pl.gx, (unsigned)pl.gy);
    if (dtype == STEP_BF16) STEP_LAUNCH((stem_wgrad16_kernel<bf16_t>), grid, dim3(SW16_NT), stream, p);
    else STEP_LAUNCH((stem_wgrad16_kernel<f16_t>), grid, dim3(SW16_NT), stream, p);
    const long long groups = (long long)pl.gy * 35 * 4 * 64;
    STEP_LAUNCH(stem_wgrad16_reduce_kernel, dim3(flat_grid(groups, 256)), dim3(256), stream, (const float*)ws, dw, pl.gx, pl.gy, Cout,
                accumulate);
    return STEP_LAUNCH_CHECK();
}


}  // extern "C"
