// step_amd/csrc/conv_igemm.hip -- fused conv + per-channel affine (+residual) + ReLU on
// channels-last activations for gfx950: im2col-free implicit GEMM on the matrix cores.
//
// Replaces, for the STEP hot path, what the reference obtains from cuDNN through
//   Unit3Dpy   = ConstantPad3d + Conv3d + BatchNorm3d(eval) + ReLU   (models/i3dpt.py:43-111)
//   Conv3d 1x1x1 / Conv2d 1x1, 3x3 / Linear of TwoBranchNet           (models/two_branch.py:60-111,182-200)
// as three or four separate kernels plus a pad copy and, per Inception block, a torch.cat copy
// (i3dpt.py:162).  Here one launch computes
//     y[.., y_coff + co] = act( sum_{tap,ci} x[pix+tap][x_coff + ci] * w[co][ci][tap] * scale[co] + shift[co] (+ res) )
// straight into a channel slice of the consumer's buffer.
//
// MI355X mapping
//   * activations are NDHWC, so the GEMM K axis (tap, ci) is contiguous in ci: a workgroup
//     stages one 32-channel slab of its input halo tile ([kd][TH+kh-1][TW+kw-1] pixels) into LDS
//     ONCE with coalesced 16-byte loads and every tap re-reads it from LDS at a shifted base --
//     no im2col matrix exists anywhere (the 27x input re-use of a 3x3x3 conv is served by LDS,
//     not by HBM/L2).  Padding is a load predicate.
//   * 64-wide wavefronts, 4 per workgroup; each wave owns a 32-pixel x (32*NB)-channel
//     accumulator built from v_mfma_f32_32x32x16_{bf16,f16} (fp32 accumulate).  The fp32
//     instantiation of the SAME code uses v_mfma_f32_32x32x2_f32 -- an exact fp32 FMA chain --
//     and is the parity path against the fp32 oracle.
//   * weights are pre-packed once into MFMA B-fragment order ([co/32][tap][ci/16][lane][8]), so
//     a wave's B operand is one fully coalesced 1 KiB load that stays L2 resident.
//   * LDS slab pixels are 64 B (16-bit) / 128 B (fp32) wide; 16-byte slots are XOR-swizzled by
//     the pixel index so the ds_read_b128 of the 32 pixels of a fragment spread over the banks.
#include "common.h"
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

namespace step {

constexpr int CK = 32;  // channels per LDS slab (two k16 MFMA steps)

// The packed weight layout is [Cout/32][taps_padded][Cin/16][lane][8]: an odd tap count > 1 is padded with
// one all-zero tap so that kernels can walk taps two at a time without a tail case.
__host__ __device__ constexpr int taps_padded(int ntaps) { return (ntaps > 1 && (ntaps & 1)) ? ntaps + 1 : ntaps; }

struct ConvParams {
    const void* x; const void* w; const float* scale; const float* shift; const void* res; void* y; void* y2;
    int split, y2_cstride, y2_coff;
    int N, D, H, W, Cin, Cout;
    int x_cstride, x_coff, y_cstride, y_coff, r_cstride, r_coff;
    int relu;
    int tiles_h, tiles_w, tiles_d;
    int gtd, gth, gtw;   // box of a general (TWL = 0) conv_tap tile, gtd*gth*gtw <= 256
    int gx, gy;          // logical grid: gx pixel tiles x gy channel groups (launched as a 1-D grid, see grid_coords)
    int nchunks;   // ceil(Cin / 32)
    int nchunks32; // same (the packed-weight K extent is 2*nchunks32 k16 blocks)
    int vec_epi;   // 16-byte output stores are legal (channel strides/offsets % 8 == 0, pointers 16-B aligned)
    int nblk32;    // ceil(Cout / 32)
    long long Mtot;  // N*D*H*W
};

// Launch order -> XCD.  Workgroup ids go round-robin over the 8 XCDs (each with its own L2), so with a plain 2-D grid
// the channel groups of one pixel tile -- which read the SAME activations -- and spatially adjacent tiles -- which
// share halos -- end up on different L2s and every one of them fetches its input from HBM / MALL again (PMC on the
// 3c fused 1x1x1: FETCH 178 MB against 51 MB of input, three channel groups).  The grid is therefore 1-D, padded to
// a multiple of 8, and remapped: ids that are consecutive on one XCD walk the channel groups of a tile first, then
// the neighbouring tiles.  Returns false for the padding workgroups (they exit before any barrier).
__device__ __forceinline__ bool grid_coords(const ConvParams& p, int& bx, int& by) {
    const unsigned id = blockIdx.x, G = gridDim.x;
    const unsigned L = (G & 7) ? id : (id & 7) * (G >> 3) + (id >> 3);
    if (L >= (unsigned)p.gx * (unsigned)p.gy) return false;
    bx = (int)(L / (unsigned)p.gy);
    by = (int)(L % (unsigned)p.gy);
    return true;
}

template <typename T> struct Ld16;  // 16-byte LDS / global vector of T
template <> struct Ld16<float> { typedef f32x4 type; };
template <> struct Ld16<bf16_t> { typedef u16x8 type; };
template <> struct Ld16<f16_t> { typedef u16x8 type; };

template <typename T>
__device__ __forceinline__ typename frag<T>::type lds_read_frag(const unsigned char* pix_base, int j, int khalf, int sw);
template <>
__device__ __forceinline__ f32x8 lds_read_frag<float>(const unsigned char* pix_base, int j, int khalf, int sw) {
    const int s0 = j * 4 + khalf * 2;
    f32x4 lo = *(const f32x4*)(pix_base + ((s0 ^ sw) << 4));
    f32x4 hi = *(const f32x4*)(pix_base + (((s0 + 1) ^ sw) << 4));
    f32x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return r;
}
template <>
__device__ __forceinline__ u16x8 lds_read_frag<bf16_t>(const unsigned char* pix_base, int j, int khalf, int sw) {
    return *(const u16x8*)(pix_base + (((j * 2 + khalf) ^ sw) << 4));
}
template <>
__device__ __forceinline__ u16x8 lds_read_frag<f16_t>(const unsigned char* pix_base, int j, int khalf, int sw) {
    return *(const u16x8*)(pix_base + (((j * 2 + khalf) ^ sw) << 4));
}

// 64-byte-per-pixel slab (conv_tap_kernel): 4 slots; fp32 holds 16 channels (one k16 step),
// 16-bit types 32 channels (two k16 steps)
template <typename T>
__device__ __forceinline__ typename frag<T>::type lds_read_slab64(const unsigned char* pix_base, int j, int khalf, int sw);
template <>
__device__ __forceinline__ f32x8 lds_read_slab64<float>(const unsigned char* pix_base, int, int khalf, int sw) {
    const int s0 = khalf * 2;
    f32x4 lo = *(const f32x4*)(pix_base + ((s0 ^ sw) << 4));
    f32x4 hi = *(const f32x4*)(pix_base + (((s0 + 1) ^ sw) << 4));
    f32x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return r;
}
template <>
__device__ __forceinline__ u16x8 lds_read_slab64<bf16_t>(const unsigned char* pix_base, int j, int khalf, int sw) {
    return *(const u16x8*)(pix_base + (((j * 2 + khalf) ^ sw) << 4));
}
template <>
__device__ __forceinline__ u16x8 lds_read_slab64<f16_t>(const unsigned char* pix_base, int j, int khalf, int sw) {
    return *(const u16x8*)(pix_base + (((j * 2 + khalf) ^ sw) << 4));
}
// B fragment out of the LDS copy of the packed weights (p already includes the lane offset)
template <typename T>
__device__ __forceinline__ typename frag<T>::type lds_read_bfrag(const unsigned char* p);
template <>
__device__ __forceinline__ f32x8 lds_read_bfrag<float>(const unsigned char* p) {
    f32x4 lo = *(const f32x4*)p;
    f32x4 hi = *(const f32x4*)(p + 16);
    f32x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return r;
}
template <>
__device__ __forceinline__ u16x8 lds_read_bfrag<bf16_t>(const unsigned char* p) { return *(const u16x8*)p; }
template <>
__device__ __forceinline__ u16x8 lds_read_bfrag<f16_t>(const unsigned char* p) { return *(const u16x8*)p; }

template <typename T>
__device__ __forceinline__ typename frag<T>::type load_b_frag(const T* p) {  // p -> this lane's 8 elements
    return *(const typename frag<T>::type*)p;
}

// TWL: log2(tile width); tile = (128 >> TWL) rows x (1 << TWL) cols of output pixels in one (n, d)
// plane.  FLAT (1x1x1 only): the tile is 128 consecutive pixels of the flattened N*D*H*W axis.
// CKT: channels per LDS slab (32, or 128 for pointwise convs with a deep Cin: 4x fewer barriers per K).
template <typename T, int TWL, int NB, int KD, int KH, int KW, bool FLAT, int CKT>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvParams p) {
    constexpr int TW = 1 << TWL, TH = 128 >> TWL;
    constexpr int HH_ = TH + KH - 1, HW_ = TW + KW - 1;
    constexpr int NPIX = FLAT ? 128 : KD * HH_ * HW_;
    constexpr int ES = (int)sizeof(T);
    constexpr int VEC = 16 / ES;
    constexpr int PITCH = CKT * ES;
    constexpr int KSTEPS = CKT / 16;
    constexpr int SLOTS = PITCH / 16;
    constexpr int PPR = (16 / SLOTS) > 0 ? (16 / SLOTS) : 1;   // pixels per 256-byte LDS row (>= 1)
    constexpr int NTAPS = KD * KH * KW;
    constexpr int NVEC = NPIX * SLOTS;
    constexpr int ITER = (NVEC + 255) / 256;
    typedef typename Ld16<T>::type vec16;
    typedef typename frag<T>::type frag_t;

    constexpr int LDSB = (NPIX * PITCH > 16384) ? NPIX * PITCH : 16384;   // >= the 16 KB epilogue transpose buffer
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDSB];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
#ifdef STEP_EMUL
    const int wave = tid >> 6;
#else
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
    const int khalf = lane >> 5;
    const int m = wave * 32 + (lane & 31);
    const int th = m >> TWL, tw = m & (TW - 1);

    // ---- which tile
    int n = 0, d = 0, h0 = 0, w0 = 0;
    long long m0 = 0;
    int gbx, gby;
    if (!grid_coords(p, gbx, gby)) return;
    if (FLAT) {
        m0 = (long long)gbx * 128;
    } else {
        int t = gbx;
        const int tw_i = t % p.tiles_w; t /= p.tiles_w;
        const int th_i = t % p.tiles_h; t /= p.tiles_h;
        d = t % p.D;
        n = t / p.D;
        h0 = th_i * TH;
        w0 = tw_i * TW;
    }
    const int nb0 = gby * NB;
    const int KC16 = p.nchunks * 2;
    const int nslab = (p.Cin + CKT - 1) / CKT;

    f32x16 acc[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    const T* xg = (const T*)p.x;
    const T* wg = (const T*)p.w;

    // this lane's B-fragment base per channel block (blocks past Cout re-read the last real block; their
    // accumulators are never stored) -- keeps the inner loop free of branches
    const T* wb[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) wb[i] = wg + ((size_t)min(nb0 + i, p.nblk32 - 1) * taps_padded(NTAPS) * KC16 * 64 + lane) * 8;

    // slab staging through registers: the loads of slab c+1 are issued as soon as slab c is in LDS and
    // fly during its MFMAs
    vec16 stage[ITER];
    auto load_slab = [&](int chunk) {
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int v = tid + it * 256;
            vec16 val;
#pragma unroll
            for (int e = 0; e < VEC; ++e) val[e] = 0;
            if (v < NVEC) {
                const int pix = v / SLOTS, slot = v % SLOTS;
                const int c = chunk * CKT + slot * VEC;
                bool inb;
                size_t gpix;
                if (FLAT) {
                    const long long gm = m0 + pix;
                    inb = gm < p.Mtot;
                    gpix = (size_t)gm;
                } else {
                    const int plane = pix / (HH_ * HW_), rem = pix % (HH_ * HW_);
                    const int r = rem / HW_, cc = rem % HW_;
                    const int id = d + plane - KD / 2, ih = h0 + r - KH / 2, iw = w0 + cc - KW / 2;
                    inb = id >= 0 && id < p.D && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
                    gpix = (((size_t)n * p.D + id) * p.H + ih) * p.W + iw;
                }
                if (inb && c < p.Cin) val = *(const vec16*)(xg + gpix * p.x_cstride + p.x_coff + c);
            }
            stage[it] = val;
        }
    };
    auto store_slab = [&]() {
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int v = tid + it * 256;
            if (v < NVEC) {
                const int pix = v / SLOTS, slot = v % SLOTS;
                const int sw = (pix / PPR) % SLOTS;
                *(vec16*)(lds + pix * PITCH + ((slot ^ sw) << 4)) = stage[it];
            }
        }
    };

    load_slab(0);
    for (int chunk = 0; chunk < nslab; ++chunk) {
        if (chunk) __syncthreads();  // all waves finished reading the previous slab
        store_slab();
        __syncthreads();
        if (chunk + 1 < nslab) load_slab(chunk + 1);

        // ---- taps x 2 k16 steps x NB accumulators
#pragma unroll 1
        for (int kd = 0; kd < KD; ++kd) {
#pragma unroll
            for (int kh = 0; kh < KH; ++kh) {
#pragma unroll
                for (int kw = 0; kw < KW; ++kw) {
                    const int tap = (kd * KH + kh) * KW + kw;
                    const int hp = FLAT ? m : ((kd * HH_ + th + kh) * HW_ + tw + kw);
                    const int sw = (hp / PPR) % SLOTS;
                    const unsigned char* pb = lds + hp * PITCH;
#pragma unroll
                    for (int j = 0; j < KSTEPS; ++j) {
                        if (chunk * KSTEPS + j >= KC16) break;          // uniform: the packed K extent ends here
                        const frag_t a = lds_read_frag<T>(pb, j, khalf, sw);
                        const size_t koff = ((size_t)tap * KC16 + chunk * KSTEPS + j) * 512;
#pragma unroll
                        for (int i = 0; i < NB; ++i) {
                            const frag_t b = load_b_frag<T>(wb[i] + koff);
                            mma_k16(a, b, acc[i], T());
                        }
                    }
                }
            }
        }
    }

    // ---- epilogue: affine (+residual) + ReLU, store channel slice(s)
    T* yg = (T*)p.y;
    const T* rg = (const T*)p.res;
    if (ES == 2 && p.vec_epi) {
        // 16-bit outputs: per 32-channel block, transpose the accumulators through LDS (fp32) so that each
        // lane stores 16 contiguous bytes (8 channels of one pixel) instead of 2.
        float* ot = (float*)lds;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            if (nb0 + i >= p.nblk32) break;                       // block-uniform
            const int cl = min((nb0 + i) * 32 + (lane & 31), p.Cout - 1);
            const float sc = p.scale ? p.scale[cl] : 1.f;
            const float sh = p.shift ? p.shift[cl] : 0.f;
            __syncthreads();                                      // LDS is free (main loop / previous block read out)
#pragma unroll
            for (int r = 0; r < 16; ++r) ot[(wave * 32 + cd_row(r, lane)) * 32 + (lane & 31)] = acc[i][r] * sc + sh;
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int idx = tid + q * 256;
                const int row = idx >> 2, g = idx & 3;
                const int co = (nb0 + i) * 32 + g * 8;
                bool ok;
                size_t opix;
                if (FLAT) {
                    const long long gm = m0 + row;
                    ok = gm < p.Mtot;
                    opix = (size_t)gm;
                } else {
                    const int oh = h0 + (row >> TWL), ow = w0 + (row & (TW - 1));
                    ok = oh < p.H && ow < p.W;
                    opix = (((size_t)n * p.D + d) * p.H + oh) * p.W + ow;
                }
                if (ok && co < p.Cout) {
                    const f32x4 lo = *(const f32x4*)(ot + row * 32 + g * 8);
                    const f32x4 hi = *(const f32x4*)(ot + row * 32 + g * 8 + 4);
                    float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    if (rg) {
                        const u16x8 rv = *(const u16x8*)(rg + opix * p.r_cstride + p.r_coff + co);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += elem<T>::from_bits16(rv[e]);
                    }
                    u16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = elem<T>::bits16(p.relu ? fmaxf(v[e], 0.f) : v[e]);
                    if (p.split > 0 && co >= p.split)
                        *(u16x8*)((T*)p.y2 + opix * p.y2_cstride + p.y2_coff + (co - p.split)) = o;
                    else
                        *(u16x8*)(yg + opix * p.y_cstride + p.y_coff + co) = o;
                }
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int co = (nb0 + i) * 32 + (lane & 31);
        if (nb0 + i < p.nblk32 && co < p.Cout) {
            const float sc = p.scale ? p.scale[co] : 1.f;
            const float sh = p.shift ? p.shift[co] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mm = wave * 32 + cd_row(r, lane);
                bool ok;
                size_t opix;
                if (FLAT) {
                    const long long gm = m0 + mm;
                    ok = gm < p.Mtot;
                    opix = (size_t)gm;
                } else {
                    const int oh = h0 + (mm >> TWL), ow = w0 + (mm & (TW - 1));
                    ok = oh < p.H && ow < p.W;
                    opix = (((size_t)n * p.D + d) * p.H + oh) * p.W + ow;
                }
                if (ok) {
                    float v = acc[i][r] * sc + sh;
                    if (rg) v += elem<T>::to_f32(rg[opix * p.r_cstride + p.r_coff + co]);
                    if (p.relu) v = fmaxf(v, 0.f);
                    if (p.split > 0 && co >= p.split)
                        ((T*)p.y2)[opix * p.y2_cstride + p.y2_coff + (co - p.split)] = elem<T>::from_f32(v);
                    else
                        yg[opix * p.y_cstride + p.y_coff + co] = elem<T>::from_f32(v);
                }
            }
        }
    }
}

// ============================================================================================
// conv_tap_kernel -- the heavy 3x3x3 / 1x3x3 path.
//
// 512 threads = 8 wavefronts own a 256-pixel x (64*NB)-channel output tile.  Waves are arranged
// 4 (pixels) x 2 (channels); each wave accumulates 2 x NB 32x32 MFMA tiles (64 px x 32*NB ch).
//   * A: a 64-byte-per-pixel slab (32 channels of 16-bit data, 16 of fp32) of the input halo tile
//     ([kd][TH+kh-1][TW+kw-1] pixels) is staged into LDS once per slab; all taps read it at shifted
//     bases (im2col-free).  Pixels sit at an 80-byte pitch (64 B + 16 B pad): consecutive pixels
//     rotate through the LDS banks without an XOR swizzle and a tap shift is a plain byte offset.
//   * B: the weights of ONE tap x slab x tile-channels (4*NB KiB, already in MFMA fragment order in
//     global memory, so the copy is linear) go global -> registers -> LDS through a double-buffered
//     LDS tile with a two-taps-deep register prefetch: the loads for tap s+2 are issued before the
//     MFMAs of tap s, the registers loaded one tap earlier are written to LDS after them, one
//     barrier per tap.  Every B fragment read from LDS feeds 2 MFMAs and every A fragment NB MFMAs
//     (a k16 step costs 2 + NB ds_read_b128 for 2*NB MFMAs), and the weights cross the L2 -> CU
//     path once per 256 pixels instead of once per 32.
// MB = 32-pixel accumulator rows per wave: MB = 2 -> 8 waves (4 x 2, 512 threads, 2 waves per SIMD);
// MB = 4 -> 4 waves (2 x 2, 256 threads, ONE wave per SIMD with the whole 512-register file: a 128-pixel x
// 96-channel wave tile reads (4 + NB) fragments per 4*NB MFMAs -- 30 % less LDS traffic per MFMA).
constexpr int CONV_GEN_NPIX = 768;         // LDS halo pixels reserved for a general (runtime-shaped) tile: 60 KiB
constexpr int CONV_GEN_NPIX_SMALL = 640;   // ... in the NB = 1 instantiation: 50 KiB + 24 KiB of weights = two workgroups
                                           // per CU (the small 14x14 / 7x7 layers are latency-bound with one)

template <typename T, int TWL, int NB, int KD, int KH, int KW, int TPS, int MB>
__global__ __launch_bounds__(MB == 2 ? 512 : 256, (NB == 1 && TWL == 0) ? 4 : 2)
void conv_tap_kernel(ConvParams p) {
    constexpr int NT = (MB == 2) ? 512 : 256;   // threads
    constexpr int WM = 8 / MB;                  // waves along the pixel axis (x 2 along channels)
    constexpr int MBP = MB / 2;                 // accumulator rows per wave per epilogue pass
    // tile shapes: TWL = 4 -> 1 plane x 16 x 16, TWL = 5 -> 1 x 8 x 32, TWL = 3 -> 4 planes x 8 x 8 (small maps:
    // 7x7 ROI features would fill 19 % of a 16x16 tile; four planes of 8x8 fill 77 %), TWL = 0 -> a box of
    // p.gtd x p.gth x p.gtw <= 256 pixels chosen at launch (GEN): power-of-two tiles cover a 28x28 map at 77 %,
    // a 50x50 one at 70 %; 2x4x28 and 2x5x25 boxes reach 88 % and 98 %.  The index arithmetic of a GEN tile uses
    // divisions, but only outside the step loop.
    constexpr bool GEN = (TWL == 0);
    constexpr int TDL = (TWL == 3) ? 2 : 0;
    constexpr int THL = GEN ? 0 : (((256 >> (TWL + TDL)) == 16) ? 4 : 3);   // log2(TH): 16 -> 4, 8 -> 3
    const int TD = GEN ? p.gtd : (1 << TDL);
    const int TW = GEN ? p.gtw : (1 << TWL), TH = GEN ? p.gth : (256 >> (TWL + TDL));
    const int TPX = TD * TH * TW;                           // pixels of the tile (256 unless GEN)
    // LDS bank conflicts of the A-fragment reads.  ds_read_b128 is serviced in four 16-lane groups ({0-3,12-15,20-27},
    // {4-11,16-19,28-31}, and the same + 32); with the 80-byte pixel pitch a group is conflict-free iff its 16 lanes
    // read pixels whose linear halo indices are distinct mod 16.  Lanes 0..31 of an MFMA row block are 32
    // consecutive tile pixels: one 32-pixel row (8x32 tile: conflict-free as is), two 16-pixel rows (16x16: row
    // pitch 18 = 2 mod 16 -> 2-way) or four 8-pixel rows (4x8x8: row pitch 10 -> 3-way; measured: 37 % of the LDS
    // cycles of the 2c layer were conflicts).  Fix: the assignment of accumulator rows to tile COLUMNS is free, so
    // odd rows of the 16x16 tile are rotated by 2 columns, and the 4x8x8 tile gets a 12-pixel row pitch plus a
    // swap of the column halves on rows 1, 2 (mod 4); the epilogue applies the same map (tile_col).
    constexpr int HWPAD = (TWL == 3) ? 2 : 0;
    const int HH_ = TH + KH - 1, HWV = TW + KW - 1, HW_ = HWV + HWPAD;   // HWV: columns that hold data
    const int PD = TD + KD - 1;                             // input planes under the tile
    const int NPIX = PD * HH_ * HW_;
    constexpr int NPIX_MAX = GEN ? (NB == 1 ? CONV_GEN_NPIX_SMALL : CONV_GEN_NPIX) : ((1 << TDL) + KD - 1) * ((256 >> (TWL + TDL)) + KH - 1) * ((1 << TWL) + KW - 1 + HWPAD);
    constexpr int ES = (int)sizeof(T);
    constexpr int VEC = 16 / ES;
    constexpr int CKT = 64 / ES;          // channels per slab: 32 (16-bit) / 16 (fp32)
    constexpr int KS = CKT / 16;          // k16 steps per slab
    constexpr int PITCH = 80, SLOTS = 4;
    constexpr int NTAPS = KD * KH * KW;
    const int NVEC = NPIX * SLOTS;
    constexpr int ITER = (NPIX_MAX * SLOTS + NT - 1) / NT;
    constexpr int FRAGB = 512 * ES;       // bytes of one B fragment (64 lanes x 8 elements)
    constexpr int FRAGV = FRAGB / 16;     // 16-byte vectors per fragment
    constexpr int NBT = 2 * NB;           // 32-channel blocks per workgroup tile
    constexpr int BTILE = NBT * KS * FRAGB;
    constexpr int BVEC = BTILE / 16;      // 16-byte vectors per tap tile
    constexpr int Q = (BVEC + NT - 1) / NT; // vectors per thread per tap
    constexpr int NTP = taps_padded(NTAPS);                 // packed taps (odd counts carry one zero tap)
    constexpr int SPS = (TPS == 1) ? NTAPS : NTP / TPS;     // pipeline steps per slab (TPS taps per barrier)
    constexpr int BSTEP = TPS * BTILE;                      // LDS weight bytes per step
    typedef typename Ld16<T>::type vec16;
    typedef typename frag<T>::type frag_t;

    __shared__ __attribute__((aligned(16))) unsigned char lds[NPIX_MAX * PITCH + 3 * BSTEP];
    unsigned char* const ldsA = lds;
    unsigned char* const ldsB = lds + NPIX_MAX * PITCH;
    // tile pixel index m (accumulator row) -> box coordinates; rows past the box of a GEN tile alias pixel 0
    auto tile_pix = [&](int m, int& td, int& th, int& tw) {
        if (GEN) {
            const int mc = m < TPX ? m : 0;
            tw = mc % TW; const int q = mc / TW; th = q % TH; td = q / TH;
        } else {
            td = m >> (TWL + THL); th = (m >> TWL) & (TH - 1); tw = m & (TW - 1);
        }
    };

    auto tile_col = [](int th, int j) {                    // tile row th, accumulator-row column slot j -> tile column
        if (TWL == 4) return (th & 1) ? ((j + 14) & 15) : j;
        if (TWL == 3) return (((th & 3) == 1) || ((th & 3) == 2)) ? (j ^ 4) : j;
        return j;
    };
    const int tid = threadIdx.x;
    const int lane = tid & 63;
#ifdef STEP_EMUL
    const int wave = tid >> 6;
#else
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform -> scalar branches
#endif
    const int khalf = lane >> 5;
    const int wm = wave % WM, wn = wave / WM;

    int gbx, gby;
    if (!grid_coords(p, gbx, gby)) return;
    int t = gbx;
    const int tw_i = t % p.tiles_w; t /= p.tiles_w;
    const int th_i = t % p.tiles_h; t /= p.tiles_h;
    const int d0 = (t % p.tiles_d) * TD;
    const int n = t / p.tiles_d;
    const int h0 = th_i * TH, w0 = tw_i * TW;
    const int HHW = HH_ * HW_;
    const int nb0 = gby * NBT;
    const int KC16 = p.nchunks32 * 2;
    const int nslab = (p.Cin + CKT - 1) / CKT;
    const int S = nslab * SPS;            // pipeline steps

    const T* xg = (const T*)p.x;
    const unsigned char* wg = (const unsigned char*)p.w;

    // LDS address of this lane's two accumulator rows (before the tap shift), incl. its k half
    const unsigned char* abase[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int m = wm * (MB * 32) + mb * 32 + (lane & 31);
        int td_, th_, tw_;
        tile_pix(m, td_, th_, tw_);
        abase[mb] = ldsA + ((td_ * HH_ + th_) * HW_ + tile_col(th_, tw_)) * PITCH + khalf * (ES == 4 ? 32 : 16);
    }

    f32x16 acc[MB][NB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][i][r] = 0.f;

    auto stage_A = [&](int slab) {
        vec16 stage[ITER];
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int v = tid + it * NT;
            vec16 val;
#pragma unroll
            for (int e = 0; e < VEC; ++e) val[e] = 0;
            if (v < NVEC) {
                const int pix = v / SLOTS, slot = v % SLOTS;
                const int c = slab * CKT + slot * VEC;
                const int plane = pix / HHW, rem = pix % HHW;
                const int r = rem / HW_, cc = rem % HW_;
                const int id = d0 + plane - KD / 2, ih = h0 + r - KH / 2, iw = w0 + cc - KW / 2;
                const bool inb = id >= 0 && id < p.D && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W && cc < HWV;
                if (inb && c < p.Cin) {
                    const size_t gpix = (((size_t)n * p.D + id) * p.H + ih) * p.W + iw;
                    val = *(const vec16*)(xg + gpix * p.x_cstride + p.x_coff + c);
                }
            }
            stage[it] = val;
        }
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int v = tid + it * NT;
            if (v < NVEC) {
                const int pix = v / SLOTS, slot = v % SLOTS;
                *(vec16*)(ldsA + pix * PITCH + (slot << 4)) = stage[it];
            }
        }
    };

    // global -> registers: this thread's vectors of the B tile of a pipeline step.  Branch-free on
    // purpose (a predicated load makes the compiler drain vmcnt at the loop head): threads without a
    // vector re-load the last one, and channel blocks past Cout re-load the last real block -- their
    // accumulators are never stored.  The per-thread part of the address is computed ONCE; a step
    // only adds a wave-uniform byte offset ((tap * KC16 + slab * KS) fragments).
    const unsigned char* wthr[Q];
    int ldsoff[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int v = min(tid + q * NT, BVEC - 1);
        const int f = v / FRAGV, within = v % FRAGV;
        const int nbl = f / KS, ks = f % KS;
        const int nbg = min(nb0 + nbl, p.nblk32 - 1);
        wthr[q] = wg + ((size_t)nbg * taps_padded(NTAPS) * KC16 + ks) * FRAGB + within * 16;
        ldsoff[q] = (tid + q * NT < BVEC) ? (tid + q * NT) * 16 : -1;
    }
    auto load_B = [&](int slab_, int sis_, u32x4 (&r)[TPS * Q]) {         // sis_ = step index within the slab
#pragma unroll
        for (int tp = 0; tp < TPS; ++tp) {
            const size_t off = (size_t)((sis_ * TPS + tp) * KC16 + slab_ * KS) * FRAGB;      // scalar
#pragma unroll
            for (int q = 0; q < Q; ++q) r[tp * Q + q] = *(const u32x4*)(wthr[q] + off);
        }
    };
    auto store_B = [&](int bufoff, const u32x4 (&r)[TPS * Q]) {
#pragma unroll
        for (int tp = 0; tp < TPS; ++tp)
#pragma unroll
            for (int q = 0; q < Q; ++q)
                if (ldsoff[q] >= 0) *(u32x4*)(ldsB + bufoff + tp * BTILE + ldsoff[q]) = r[tp * Q + q];
    };

    // Pipeline.  A STEP = TPS taps of one slab = one barrier; a SLOT = one tap.
    //   weights: TWO register sets (R0 on even steps, R1 on odd ones) and THREE LDS step-buffers.  During step s
    //     the set of that parity, which holds step s+2 (loaded during step s-2), is written to buffer (s+2)%3
    //     -- last read during step s-1 -- and re-issued for step s+4: two steps of matrix work (~1500 cycles)
    //     cover the L2 latency of the weight stream; with one set the load -> store distance was a single step.
    //   fragments: two register sets alternating per slot.  The ds_reads of slot u+1 are issued BEFORE
    //     the MFMAs of slot u (for the first slot of a step they come from the next buffer, complete
    //     since the previous barrier), so LDS latency, the weight hand-over and the barrier hide behind
    //     matrix work.  A slab switch drains the pipeline once per slab.
    u32x4 R0[TPS * Q], R1[TPS * Q];
    frag_t fa[2][KS][MB], fb[2][KS][NB];

    const unsigned char* const bwave = ldsB + (wn * NB) * KS * FRAGB + lane * (8 * ES);
    auto read_frags = [&](auto setc, int bufoff, int shift) {
        constexpr int SET = decltype(setc)::value;
#pragma unroll
        for (int j = 0; j < KS; ++j) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) fa[SET][j][mb] = lds_read_bfrag<T>(abase[mb] + shift + j * 32);
#pragma unroll
            for (int i = 0; i < NB; ++i) fb[SET][j][i] = lds_read_bfrag<T>(bwave + bufoff + (i * KS + j) * FRAGB);
        }
    };
    auto mma_all = [&](auto setc) {
        constexpr int SET = decltype(setc)::value;
#pragma unroll
        for (int j = 0; j < KS; ++j)
#pragma unroll
            for (int i = 0; i < NB; ++i) {   // channel blocks past Cout compute on a duplicate block and are never stored
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) mma_k16(fa[SET][j][mb], fb[SET][j][i], acc[mb][i], T());
            }
    };
    // LDS byte shift of tap t of the slab (the zero tap of an odd tap count reads tap 0's pixels)
    auto tap_shift = [&](int t_) {
        const int tt = (t_ < NTAPS) ? t_ : 0;
        return (((tt / (KH * KW)) * HH_ + (tt / KW) % KH) * HW_ + tt % KW) * PITCH;
    };

    // (slab, step-in-slab) cursors: c0 = current step, c3 = step + 4 (weight loads)
    int slab0 = 0, sis0 = 0, slab3 = 0, sis3 = 0;
    auto adv = [&](int& sl, int& si) { if (++si == SPS) { si = 0; ++sl; } };
    auto adv_clamped = [&]() { adv(slab3, sis3); if (slab3 >= nslab) { slab3 = nslab - 1; sis3 = SPS - 1; } };   // past the end: re-read the last tile

    stage_A(0);
    load_B(0, 0, R0);
    adv_clamped();
    load_B(slab3, sis3, R1);
    adv_clamped();
    store_B(0, R0);
    load_B(slab3, sis3, R0);                               // step 2
    adv_clamped();
    store_B(BSTEP, R1);
    load_B(slab3, sis3, R1);                               // step 3
    adv_clamped();                                         // -> step 4
    __syncthreads();
    read_frags(std::integral_constant<int, 0>(), 0, 0);

    int b0 = 0, b1 = BSTEP, b2 = 2 * BSTEP;                // LDS buffers of step s, s+1, s+2
    int s_ = 0;                                            // current step
    auto slot = [&](auto setc, auto tpc, u32x4 (&R)[TPS * Q]) {
        constexpr int SET = decltype(setc)::value;
        constexpr int TP = decltype(tpc)::value;           // tap slot within the step
        constexpr bool LAST = (TP == TPS - 1);
        bool new_slab = false;
        if (!LAST) {                                       // next slot: same step, same weight buffer
            read_frags(std::integral_constant<int, SET ^ 1>(), b0 + (TP + 1) * BTILE, tap_shift(sis0 * TPS + TP + 1));
        } else {                                           // next slot opens step s+1
            new_slab = (sis0 + 1 == SPS);
            if (s_ + 1 < S && !new_slab)
                read_frags(std::integral_constant<int, SET ^ 1>(), b1, tap_shift((sis0 + 1) * TPS));
        }
        mma_all(setc);
        if (LAST) {
            store_B(b2, R);                                // (past the end: a duplicate tile into a buffer nobody reads)
            load_B(slab3, sis3, R);
            adv_clamped();
            if (s_ + 1 < S && new_slab) {
                __syncthreads();                          // every wave is done with this slab of A
                stage_A(slab0 + 1);
                __syncthreads();
                read_frags(std::integral_constant<int, SET ^ 1>(), b1, 0);
            }
            __syncthreads();
            const int t0 = b0; b0 = b1; b1 = b2; b2 = t0;  // rotate the three buffers
            adv(slab0, sis0);
            ++s_;
        }
    };
    if (TPS == 2) {
#pragma unroll 1
        while (s_ < S) {
            slot(std::integral_constant<int, 0>(), std::integral_constant<int, 0>(), R0);
            slot(std::integral_constant<int, 1>(), std::integral_constant<int, TPS - 1>(), R0);
            if (s_ < S) {
                slot(std::integral_constant<int, 0>(), std::integral_constant<int, 0>(), R1);
                slot(std::integral_constant<int, 1>(), std::integral_constant<int, TPS - 1>(), R1);
            }
        }
    } else {
#pragma unroll 1
        while (s_ < S) {
            slot(std::integral_constant<int, 0>(), std::integral_constant<int, TPS - 1>(), R0);
            if (s_ < S) slot(std::integral_constant<int, 1>(), std::integral_constant<int, TPS - 1>(), R1);
        }
    }

    // ---- epilogue
    T* yg = (T*)p.y;
    const T* rg = (const T*)p.res;
    if (ES == 2 && p.vec_epi) {
        // 16-bit outputs: transpose the accumulators through LDS (fp32, 128 pixels at a time) so that
        // every lane stores 16 contiguous bytes (8 channels of one pixel): 8x fewer store instructions
        // than the accumulator layout allows (2 B per lane, 64 B runs) and whole-line writes.
        constexpr int BN = NBT * 32, G = BN / 8;
        float* ot = (float*)lds;
        float sc[NB], sh[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int co = min((nb0 + wn * NB + i) * 32 + (lane & 31), p.Cout - 1);
            sc[i] = p.scale ? p.scale[co] : 1.f;
            sh[i] = p.shift ? p.shift[co] : 0.f;
        }
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {                  // two passes of 128 pixels
            if (ps) __syncthreads();                      // previous half has been read out
#pragma unroll
            for (int mbl = 0; mbl < MBP; ++mbl)
#pragma unroll
                for (int i = 0; i < NB; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        ot[((wm * MBP + mbl) * 32 + cd_row(r, lane)) * BN + (wn * NB + i) * 32 + (lane & 31)] =
                            acc[ps * MBP + mbl][i][r] * sc[i] + sh[i];
            __syncthreads();
            for (int idx = tid; idx < 128 * G; idx += NT) {
                const int row = idx / G, g = idx % G;
                // row = (wm*MBP + mbl)*32 + rr  ->  tile pixel wm*(MB*32) + (ps*MBP + mbl)*32 + rr
                const int mm = ((row >> 5) / MBP) * (MB * 32) + (ps * MBP + (row >> 5) % MBP) * 32 + (row & 31);
                int tdl, thl, twl;
                tile_pix(mm, tdl, thl, twl);
                const int od = d0 + tdl, oh = h0 + thl, ow = w0 + tile_col(thl, twl);
                const int co = nb0 * 32 + g * 8;
                if ((!GEN || mm < TPX) && od < p.D && oh < p.H && ow < p.W && co < p.Cout) {
                    const size_t opix = (((size_t)n * p.D + od) * p.H + oh) * p.W + ow;
                    const f32x4 lo = *(const f32x4*)(ot + row * BN + g * 8);
                    const f32x4 hi = *(const f32x4*)(ot + row * BN + g * 8 + 4);
                    float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    if (rg) {
                        const u16x8 rv = *(const u16x8*)(rg + opix * p.r_cstride + p.r_coff + co);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += elem<T>::from_bits16(rv[e]);
                    }
                    u16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = elem<T>::bits16(p.relu ? fmaxf(v[e], 0.f) : v[e]);
                    *(u16x8*)(yg + opix * p.y_cstride + p.y_coff + co) = o;
                }
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int nbg = nb0 + wn * NB + i;
        const int co = nbg * 32 + (lane & 31);
        if (nbg < p.nblk32 && co < p.Cout) {
            const float sc = p.scale ? p.scale[co] : 1.f;
            const float sh = p.shift ? p.shift[co] : 0.f;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int mm = wm * (MB * 32) + mb * 32 + cd_row(r, lane);
                    int tdl, thl, twl;
                    tile_pix(mm, tdl, thl, twl);
                    const int od = d0 + tdl, oh = h0 + thl, ow = w0 + tile_col(thl, twl);
                    if ((!GEN || mm < TPX) && od < p.D && oh < p.H && ow < p.W) {
                        const size_t opix = (((size_t)n * p.D + od) * p.H + oh) * p.W + ow;
                        float v = acc[mb][i][r] * sc + sh;
                        if (rg) v += elem<T>::to_f32(rg[opix * p.r_cstride + p.r_coff + co]);
                        if (p.relu) v = fmaxf(v, 0.f);
                        yg[opix * p.y_cstride + p.y_coff + co] = elem<T>::from_f32(v);
                    }
                }
            }
        }
    }
}

// ============================================================================================
// conv_pw_kernel -- pointwise (1x1x1) convs / Linear layers with a deep K: a streaming GEMM.
// 512 threads = 8 wavefronts (4 x 2) own 256 consecutive pixels x (64*NB) channels; each wave a
// 64-pixel x (32*NB)-channel block.  One pipeline step = one 64-byte slab of input channels (32 x
// 16-bit / 16 x fp32).  BOTH operands stream: the A slab (256 pixels x 64 B, 80-byte pitch) and the
// weight tile go global -> one register set each -> three-buffer LDS rings; fragments are double-
// buffered in registers, so the ds_reads of step s+1 are issued before the MFMAs of step s and there
// is one barrier per step (the conv_tap_kernel pipeline without a resident halo tile).  Loads are
// branch-free (clamped addresses + bit masks) so the compiler keeps exact vmcnt waits in the loop.
template <typename T, int NB>
__global__ __launch_bounds__(512) void conv_pw_kernel(ConvParams p) {
    constexpr int ES = (int)sizeof(T);
    constexpr int VEC = 16 / ES;
    constexpr int CKT = 64 / ES, KS = CKT / 16;
    constexpr int PITCH = 80;
    constexpr int ATILE = 256 * PITCH;                       // 20480 B
    constexpr int FRAGB = 512 * ES, FRAGV = FRAGB / 16;
    constexpr int NBT = 2 * NB;
    constexpr int BTILE = NBT * KS * FRAGB;
    constexpr int BVEC = BTILE / 16;
    constexpr int Q = (BVEC + 511) / 512;
    typedef typename frag<T>::type frag_t;

    constexpr int BSTRIDE = Q * 512 * 16;                    // weight buffer pitch: every thread stores all its Q vectors (no predicate)
    __shared__ __attribute__((aligned(16))) unsigned char lds[3 * ATILE + 3 * BSTRIDE];
    unsigned char* const ldsA = lds;
    unsigned char* const ldsB = lds + 3 * ATILE;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
#ifdef STEP_EMUL
    const int wave = tid >> 6;
#else
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
    const int khalf = lane >> 5;
    const int wm = wave & 3, wn = wave >> 2;
    int gbx, gby;
    if (!grid_coords(p, gbx, gby)) return;
    const long long m0 = (long long)gbx * 256;
    const int nb0 = gby * NBT;
    const int KC16 = p.nchunks32 * 2;
    const int S = (p.Cin + CKT - 1) / CKT;

    const unsigned char* xg = (const unsigned char*)p.x;
    const unsigned char* wg = (const unsigned char*)p.w;

    // A: two 16-byte vectors per thread per step (pixel = v / 4, slot = v % 4)
    const unsigned char* athr[2];
    unsigned int amask[2];
    int acol[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int v = tid + q * 512;
        const int pix = v >> 2, slot = v & 3;
        const long long gm = m0 + pix;
        const bool ok = gm < p.Mtot;
        athr[q] = xg + ((size_t)(ok ? gm : 0) * p.x_cstride + p.x_coff) * ES;
        amask[q] = ok ? 0xffffffffu : 0u;
        acol[q] = slot * VEC;
    }
    // B: this thread's vectors of a step tile (as conv_tap_kernel)
    const unsigned char* wthr[Q];
    int ldsoff[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int v = min(tid + q * 512, BVEC - 1);
        const int f = v / FRAGV, within = v % FRAGV;
        const int nbl = f / KS, ks = f % KS;
        const int nbg = min(nb0 + nbl, p.nblk32 - 1);
        wthr[q] = wg + ((size_t)nbg * KC16 + ks) * FRAGB + within * 16;
        ldsoff[q] = (tid + q * 512) * 16;
    }
    // global -> register ring of DR step slabs -> LDS ring of 3: a slab is loaded DR steps before it is written to
    // LDS (4 steps of matrix work cover the HBM latency; with one register set the load -> store distance was a
    // single step and the K loop ran latency-bound)
    constexpr int DR = 4;
    u32x4 RA[DR][2], RB[DR][Q];
    auto load_step = [&](auto rc, int s_) {
        constexpr int RS = decltype(rc)::value;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int c = s_ * CKT + acol[q];
            const bool cok = c < p.Cin;                                   // whole vector in or out (Cin % VEC == 0)
            RA[RS][q] = *(const u32x4*)(athr[q] + (size_t)(cok ? c : 0) * ES);     // masked when it is written to LDS: an
                                                                                    // AND here would wait for the load at once
        }
        const size_t off = (size_t)(min(s_, S - 1) * KS) * FRAGB;       // past the end: a harmless re-read of the last tile
#pragma unroll
        for (int q = 0; q < Q; ++q) RB[RS][q] = *(const u32x4*)(wthr[q] + off);
    };
    auto store_step = [&](auto rc, int buf, int slab) {
        constexpr int RS = decltype(rc)::value;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int v = tid + q * 512;
            const unsigned int mk = (slab * CKT + acol[q] < p.Cin) ? amask[q] : 0u;
            *(u32x4*)(ldsA + buf * ATILE + (v >> 2) * PITCH + ((v & 3) << 4)) = RA[RS][q] & mk;
        }
#pragma unroll
        for (int q = 0; q < Q; ++q)
            *(u32x4*)(ldsB + buf * BSTRIDE + ldsoff[q]) = RB[RS][q];
    };

    const unsigned char* abase[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) abase[mb] = ldsA + (wm * 64 + mb * 32 + (lane & 31)) * PITCH + khalf * (ES == 4 ? 32 : 16);
    const unsigned char* const bwave = ldsB + (wn * NB) * KS * FRAGB + lane * (8 * ES);

    f32x16 acc[2][NB];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][i][r] = 0.f;

    frag_t fa[2][KS][2], fb[2][KS][NB];
    auto read_frags = [&](auto setc, int buf) {
        constexpr int SET = decltype(setc)::value;
#pragma unroll
        for (int j = 0; j < KS; ++j) {
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) fa[SET][j][mb] = lds_read_bfrag<T>(abase[mb] + buf * ATILE + j * 32);
#pragma unroll
            for (int i = 0; i < NB; ++i) fb[SET][j][i] = lds_read_bfrag<T>(bwave + buf * BSTRIDE + (i * KS + j) * FRAGB);
        }
    };
    auto mma_all = [&](auto setc) {
        constexpr int SET = decltype(setc)::value;
#pragma unroll
        for (int j = 0; j < KS; ++j)
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                mma_k16(fa[SET][j][0], fb[SET][j][i], acc[0][i], T());
                mma_k16(fa[SET][j][1], fb[SET][j][i], acc[1][i], T());
            }
    };

    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    typedef std::integral_constant<int, 2> I2;
    typedef std::integral_constant<int, 3> I3;
    // prologue: steps 0..3 in flight at once, 0 and 1 to LDS, 4 and 5 take their register sets
    load_step(I0(), 0); load_step(I1(), 1); load_step(I2(), 2); load_step(I3(), 3);
    store_step(I0(), 0, 0);
    store_step(I1(), 1, 1);
    load_step(I0(), 4); load_step(I1(), 5);
    __syncthreads();
    read_frags(I0(), 0);

    int b1 = 1, b2 = 2, s_ = 0;
    // step s: register set (s + 2) & 3 holds slab s + 2 -> LDS buffer (s + 2) % 3, then reloads slab s + 6.
    // No predicates inside (loads past the end are clamped and masked, the surplus fragment read hits a valid
    // buffer): any branch in the loop makes the compiler fall back to vmcnt(0) waits.
    auto step = [&](auto setc, auto rc) {
        constexpr int SET = decltype(setc)::value;
        read_frags(std::integral_constant<int, SET ^ 1>(), b1);
        mma_all(setc);
        store_step(rc, b2, s_ + 2);
        load_step(rc, s_ + 6);
        __syncthreads();
        const int nb = (b2 == 2) ? 0 : b2 + 1;
        b1 = b2; b2 = nb;
        ++s_;
    };
#pragma unroll 1
    while (s_ + 4 <= S) {
        step(I0(), I2());
        step(I1(), I3());
        step(I0(), I0());
        step(I1(), I1());
    }
    if (s_ < S) {                                            // 1..3 remaining steps
        step(I0(), I2());
        if (s_ < S) step(I1(), I3());
        if (s_ < S) step(I0(), I0());
    }

    // ---- epilogue (two destinations supported)
    T* yg = (T*)p.y;
    const T* rg = (const T*)p.res;
    if (ES == 2 && p.vec_epi) {
        constexpr int BN = NBT * 32, G = BN / 8;
        float* ot = (float*)lds;
        float sc[NB], sh[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int co = min((nb0 + wn * NB + i) * 32 + (lane & 31), p.Cout - 1);
            sc[i] = p.scale ? p.scale[co] : 1.f;
            sh[i] = p.shift ? p.shift[co] : 0.f;
        }
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            if (mb) __syncthreads();
#pragma unroll
            for (int i = 0; i < NB; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    ot[(wm * 32 + cd_row(r, lane)) * BN + (wn * NB + i) * 32 + (lane & 31)] = acc[mb][i][r] * sc[i] + sh[i];
            __syncthreads();
            for (int idx = tid; idx < 128 * G; idx += 512) {
                const int row = idx / G, g = idx % G;
                const long long gm = m0 + (row >> 5) * 64 + mb * 32 + (row & 31);
                const int co = nb0 * 32 + g * 8;
                if (gm < p.Mtot && co < p.Cout) {
                    const size_t opix = (size_t)gm;
                    const f32x4 lo = *(const f32x4*)(ot + row * BN + g * 8);
                    const f32x4 hi = *(const f32x4*)(ot + row * BN + g * 8 + 4);
                    float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    if (rg) {
                        const u16x8 rv = *(const u16x8*)(rg + opix * p.r_cstride + p.r_coff + co);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += elem<T>::from_bits16(rv[e]);
                    }
                    u16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = elem<T>::bits16(p.relu ? fmaxf(v[e], 0.f) : v[e]);
                    if (p.split > 0 && co >= p.split)
                        *(u16x8*)((T*)p.y2 + opix * p.y2_cstride + p.y2_coff + (co - p.split)) = o;
                    else
                        *(u16x8*)(yg + opix * p.y_cstride + p.y_coff + co) = o;
                }
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int nbg = nb0 + wn * NB + i;
        const int co = nbg * 32 + (lane & 31);
        if (nbg < p.nblk32 && co < p.Cout) {
            const float sc = p.scale ? p.scale[co] : 1.f;
            const float sh = p.shift ? p.shift[co] : 0.f;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long long gm = m0 + wm * 64 + mb * 32 + cd_row(r, lane);
                    if (gm < p.Mtot) {
                        const size_t opix = (size_t)gm;
                        float v = acc[mb][i][r] * sc + sh;
                        if (rg) v += elem<T>::to_f32(rg[opix * p.r_cstride + p.r_coff + co]);
                        if (p.relu) v = fmaxf(v, 0.f);
                        if (p.split > 0 && co >= p.split)
                            ((T*)p.y2)[opix * p.y2_cstride + p.y2_coff + (co - p.split)] = elem<T>::from_f32(v);
                        else
                            yg[opix * p.y_cstride + p.y_coff + co] = elem<T>::from_f32(v);
                    }
                }
            }
        }
    }
}

// ============================================================================================
// pw_splitk_kernel -- pointwise convs / Linear layers with FEW rows and a very deep K (the heads'
// Linear(12544 -> 60 / 12) on N*Tl <= a few hundred rows: two_branch.py:196,209-211).  As a tiled GEMM this is
// 2-8 workgroups walking 98 slabs one after the other (measured 228 us per call); it is weight- and
// activation-bandwidth work that wants the whole chip.  Here K is split across workgroups:
//   grid = (32-channel block, K chunk, 128-row tile); 4 waves per workgroup take the chunk's k16 steps
//   round-robin, reading A fragments straight from global memory (16 B per lane, the MFMA operand layout)
//   and B fragments from the fragment-ordered packed weights (1 KiB per wave, fully coalesced): no LDS and
//   no barrier in the loop.  The four waves' accumulators are summed through LDS and written as one fp32
//   partial tile to ws[chunk][row][channel]; pw_splitk_finish_kernel sums the chunks in a fixed order
//   (deterministic, no atomics), applies affine / residual / ReLU and stores in the storage type.
template <typename T, int MBK>
__global__ __launch_bounds__(256) void pw_splitk_kernel(ConvParams p, float* __restrict__ ws, int kchunk16, int mpad, int cpad) {
    static_assert(sizeof(T) == 2, "16-bit storage types only");
    typedef typename frag<T>::type frag_t;
    __shared__ float red[4][MBK * 32][33];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, khalf = lane >> 5;
    const int nb = blockIdx.x;
    const long long m0 = (long long)blockIdx.z * (MBK * 32);
    const int KC16 = p.nchunks32 * 2;
    const int ks_beg = blockIdx.y * kchunk16, ks_end = min(ks_beg + kchunk16, KC16);
    const T* xg = (const T*)p.x;
    const T* wg = (const T*)p.w + ((size_t)nb * KC16 * 64 + lane) * 8;

    const T* arow[MBK];
    bool rok[MBK];
#pragma unroll
    for (int mb = 0; mb < MBK; ++mb) {
        const long long r = m0 + mb * 32 + (lane & 31);
        rok[mb] = r < p.Mtot;
        arow[mb] = xg + (size_t)(rok[mb] ? r : 0) * p.x_cstride + p.x_coff + khalf * 8;
    }
    f32x16 acc[MBK];
#pragma unroll
    for (int mb = 0; mb < MBK; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;

    for (int ks = ks_beg + wave; ks < ks_end; ks += 4) {
        const frag_t b = load_b_frag<T>(wg + (size_t)ks * 512);
        const int c = ks * 16 + khalf * 8;
        const bool cok = c < p.Cin;                        // Cin % 8 == 0: a fragment half is all in or all out
#pragma unroll
        for (int mb = 0; mb < MBK; ++mb) {
            frag_t a;
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] = 0;
            if (cok && rok[mb]) a = *(const frag_t*)(arow[mb] + ks * 16);
            mma_k16(a, b, acc[mb], T());
        }
    }
#pragma unroll
    for (int mb = 0; mb < MBK; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave][mb * 32 + cd_row(r, lane)][lane & 31] = acc[mb][r];
    __syncthreads();
    float* out = ws + ((size_t)blockIdx.y * mpad + m0) * cpad + nb * 32;
    for (int idx = tid; idx < MBK * 32 * 32; idx += 256) {
        const int row = idx >> 5, col = idx & 31;
        if (m0 + row < mpad)
            out[(size_t)row * cpad + col] = (red[0][row][col] + red[1][row][col]) + (red[2][row][col] + red[3][row][col]);
    }
}

template <typename T>
__global__ void pw_splitk_finish_kernel(ConvParams p, const float* __restrict__ ws, int ksplit, int mpad, int cpad) {
    const long long total = p.Mtot * p.Cout;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)blockDim.x * gridDim.x) {
        const long long row = idx / p.Cout;
        const int co = (int)(idx % p.Cout);
        float v = 0.f;
        for (int k = 0; k < ksplit; ++k) v += ws[((size_t)k * mpad + row) * cpad + co];
        v = v * (p.scale ? p.scale[co] : 1.f) + (p.shift ? p.shift[co] : 0.f);
        if (p.res) v += elem<T>::to_f32(((const T*)p.res)[(size_t)row * p.r_cstride + p.r_coff + co]);
        if (p.relu) v = fmaxf(v, 0.f);
        if (p.split > 0 && co >= p.split) ((T*)p.y2)[(size_t)row * p.y2_cstride + p.y2_coff + (co - p.split)] = elem<T>::from_f32(v);
        else ((T*)p.y)[(size_t)row * p.y_cstride + p.y_coff + co] = elem<T>::from_f32(v);
    }
}

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
    const void* x; const float* dy; float* dw;
    int N, D, H, W, Cin, Cout, kd, kh, kw;
    int x_cstride, x_coff, dy_cstride, dy_coff;
    int cot, cit;                 // tiles along Cout / Cin
    int rows;                     // (n, d, h) rows per wavefront job
    long long total_rows;         // N * D * H
    long long jobs;               // ceil(total_rows / rows)
};

template <typename T, int MB, int NB>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradParams p) {
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
        const float* dyrow = p.dy + ((((size_t)n * p.D + d) * p.H + h) * p.W) * p.dy_cstride + p.dy_coff;
        const T* xrow = (const T*)p.x + ((((size_t)n * p.D + id) * p.H + ih) * p.W) * p.x_cstride + p.x_coff;
        for (int w0 = 0; w0 < p.W; w0 += 16) {
            f32x8 a[MB], b[NB];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int w = w0 + 8 * khalf + j, iw = w + kw_ - p.kw / 2;
                const bool aok = w < p.W, bok = aok && iw >= 0 && iw < p.W;
                const int wc = aok ? w : p.W - 1, iwc = bok ? iw : 0;
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    const float v = dyrow[(size_t)wc * p.dy_cstride + coc[mb]];
                    a[mb][j] = (aok && cook[mb]) ? v : 0.f;
                }
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const float v = elem<T>::to_f32(xrow[(size_t)iwc * p.x_cstride + cic[nb]]);
                    b[nb][j] = (bok && ciok[nb]) ? v : 0.f;
                }
            }
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) mma_k16(a[mb], b[nb], acc[mb][nb], float());
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
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int ow = w0 + 8 * khalf + j, iw = 2 * ow + kw_ - 2;
                const bool aok = ow < p.Wo, bok = aok && nok && iw >= 0 && iw < p.W;
                const int owc = aok ? ow : p.Wo - 1, iwc = bok ? iw : 0;
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
                    const float v = dyrow[(size_t)owc * p.Cout + coc[mb]];
                    a[mb][j] = (aok && cook[mb]) ? v : 0.f;
                }
                const float xv = elem<T>::to_f32(xrow[iwc]);
                b[j] = bok ? xv : 0.f;
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

// ---- weight packing: torch [Cout][Cin][taps] fp32 -> [nb32][tap][kc16][lane][8] of T ----------
template <typename T>
__global__ void pack_weight_kernel(const float* __restrict__ w, const int32_t* __restrict__ perm, T* __restrict__ out,
                                   int Cout, int Cin, int ntaps, int KC16, long long total) {
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)blockDim.x * gridDim.x) {
        const int e = (int)(idx & 7);
        const int lane = (int)((idx >> 3) & 63);
        long long q = idx >> 9;
        const int kc16 = (int)(q % KC16); q /= KC16;
        const int ntp = taps_padded(ntaps);
        const int tap = (int)(q % ntp);
        const int nb = (int)(q / ntp);
        const int co = nb * 32 + (lane & 31);
        const int ci = kc16 * 16 + (lane >> 5) * 8 + e;
        float v = 0.f;
        if (co < Cout && ci < Cin && tap < ntaps) {
            const int cs = perm ? perm[ci] : ci;
            v = w[((size_t)co * Cin + cs) * ntaps + tap];
        }
        out[idx] = elem<T>::from_f32(v);
    }
}

// ============================================================================================
// The I3D stem: 7x7x7, stride 2, Cin = 3, pad (2 front, 3 back) + affine + ReLU.
// Input in the reference's own layout x[N][T][3][H][W]; output channels-last.
// The slab in LDS is [7 frames][2*TH+5 rows][40 cols] pixels of 4 channels (c = 3 is zero) and
// the GEMM K axis is ordered (kd, kh, kw(8, the 8th tap has zero weight), c(4)): the 8 taps x 4
// channels an output pixel needs from one input row are 32 CONTIGUOUS, 16-byte aligned
// elements, so the stride-2 gather is again a plain ds_read_b128 per lane.  K = 7*7*32 = 1568.
constexpr int STEM_TH = 8, STEM_TW = 16;
constexpr int STEM_ROWS = 2 * STEM_TH + 5, STEM_COLS = 40;

struct StemParams {
    const void* x; const void* w; const float* scale; const float* shift; void* y;
    int N, T, H, W, To, Ho, Wo, Cout, y_cstride, y_coff;
    int tiles_h, tiles_w, nblk32;
};

template <typename T, int NB>
__global__ __launch_bounds__(256) void stem_igemm_kernel(StemParams p) {
    constexpr int ES = (int)sizeof(T);
    constexpr int PIXB = 4 * ES;  // bytes per LDS pixel (4 channels)
    constexpr int NPIX = 7 * STEM_ROWS * STEM_COLS;
    typedef typename frag<T>::type frag_t;
    __shared__ __attribute__((aligned(16))) unsigned char lds[NPIX * PIXB];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int khalf = lane >> 5;
    const int m = wave * 32 + (lane & 31);
    const int th = m >> 4, tw = m & 15;

    int t = blockIdx.x;
    const int tw_i = t % p.tiles_w; t /= p.tiles_w;
    const int th_i = t % p.tiles_h; t /= p.tiles_h;
    const int od = t % p.To;
    const int n = t / p.To;
    const int oh0 = th_i * STEM_TH, ow0 = tw_i * STEM_TW;
    const int nb0 = blockIdx.y * NB;

    // ---- stage: LDS col cl <-> input col iw = 2*ow0 - 4 + cl ; row r <-> ih = 2*oh0 - 2 + r ;
    //      frame f <-> it = 2*od - 2 + f.  Items = 4 consecutive cols of one (frame,row).
    const T* xg = (const T*)p.x;
    const bool vec_ok = (p.W % 4) == 0;
    for (int item = tid; item < 7 * STEM_ROWS * (STEM_COLS / 4); item += 256) {
        const int cq = item % (STEM_COLS / 4);
        const int r = (item / (STEM_COLS / 4)) % STEM_ROWS;
        const int f = item / ((STEM_COLS / 4) * STEM_ROWS);
        const int it = 2 * od - 2 + f, ih = 2 * oh0 - 2 + r, iw0 = 2 * ow0 - 4 + cq * 4;
        T px[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int c = 0; c < 4; ++c) px[a][c] = elem<T>::from_f32(0.f);
        if (it >= 0 && it < p.T && ih >= 0 && ih < p.H) {
            const size_t base = (((size_t)n * p.T + it) * 3) * p.H * p.W + (size_t)ih * p.W;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const T* src = xg + base + (size_t)c * p.H * p.W;
                if (vec_ok && iw0 >= 0 && iw0 + 3 < p.W) {
                    typedef unsigned int uvec __attribute__((ext_vector_type(ES)));   // 4 elements = 2*ES... bytes
                    uvec raw = *(const uvec*)(src + iw0);
                    T v4[4];
                    __builtin_memcpy(v4, &raw, sizeof(v4));
#pragma unroll
                    for (int a = 0; a < 4; ++a) px[a][c] = v4[a];
                } else {
#pragma unroll
                    for (int a = 0; a < 4; ++a)
                        if (iw0 + a >= 0 && iw0 + a < p.W) px[a][c] = src[iw0 + a];
                }
            }
        }
        unsigned char* dst = lds + ((f * STEM_ROWS + r) * STEM_COLS + cq * 4) * PIXB;
#pragma unroll
        for (int q = 0; q < (4 * PIXB) / 16; ++q) {
            u32x4 tmp;
            __builtin_memcpy(&tmp, (const unsigned char*)&px[0][0] + 16 * q, 16);
            *(u32x4*)(dst + 16 * q) = tmp;
        }
    }
    __syncthreads();

    f32x16 acc[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    const T* wg = (const T*)p.w;
#pragma unroll 1
    for (int kd = 0; kd < 7; ++kd) {
#pragma unroll 1
        for (int kh = 0; kh < 7; ++kh) {
            // pixel (2*tw + 2 + 0) of row (2*th + kh) of frame kd; this lane's 8 elements of step j
            // start at tap kw = 4*j + 2*khalf
            const unsigned char* rowb = lds + ((kd * STEM_ROWS + 2 * th + kh) * STEM_COLS + 2 * tw + 2) * PIXB;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const unsigned char* ap = rowb + (4 * j + 2 * khalf) * PIXB;
                frag_t a;
                {
                    u32x4 h2[ES / 2];
#pragma unroll
                    for (int q = 0; q < ES / 2; ++q) h2[q] = *(const u32x4*)(ap + 16 * q);
                    __builtin_memcpy(&a, h2, sizeof(a));
                }
                const int ks = (kd * 7 + kh) * 2 + j;
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    if (nb0 + i < p.nblk32) {
                        const T* bp = wg + (((size_t)(nb0 + i) * 98 + ks) * 64 + lane) * 8;
                        const frag_t b = load_b_frag<T>(bp);
                        mma_k16(a, b, acc[i], T());
                    }
                }
            }
        }
    }

    T* yg = (T*)p.y;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int co = (nb0 + i) * 32 + (lane & 31);
        if (nb0 + i < p.nblk32 && co < p.Cout) {
            const float sc = p.scale ? p.scale[co] : 1.f;
            const float sh = p.shift ? p.shift[co] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mm = wave * 32 + cd_row(r, lane);
                const int oh = oh0 + (mm >> 4), ow = ow0 + (mm & 15);
                if (oh < p.Ho && ow < p.Wo) {
                    float v = fmaxf(acc[i][r] * sc + sh, 0.f);
                    const size_t opix = (((size_t)n * p.To + od) * p.Ho + oh) * p.Wo + ow;
                    yg[opix * p.y_cstride + p.y_coff + co] = elem<T>::from_f32(v);
                }
            }
        }
    }
}

// --------------------------------------------------------------------------------------------
// stem_tap_kernel (16-bit types): the pipelined form of the stem.
// 256 threads = 4 wavefronts own a 16x16-pixel x 64-channel output tile of one output frame; each
// wave accumulates a 64-pixel x 64-channel block (2 x 2 MFMA tiles).  One pipeline step = one
// (kd, kh) pair = 32 K values (8 kw taps x 4 channels), 49 steps.
//   * input frames go through a 3-slot LDS ring ([37 rows][40 cols] pixels of 4 channels each): frame
//     kd+2 is loaded into registers at the first step of frame kd and written to the slot frame kd-1
//     left, so only 35 KB of LDS hold the 7-frame receptive field and 3 workgroups fit on a CU
//     (their staging bubbles fill each other's matrix work);
//   * weights: one register set + 3 LDS buffers, fragments: 2 register sets (as conv_tap_kernel);
//   * epilogue: LDS transpose, 16-byte stores.
constexpr int STP_ROWS = 37, STP_COLS = 40;

template <typename T>
__global__ __launch_bounds__(256) void stem_tap_kernel(StemParams p) {
    static_assert(sizeof(T) == 2, "16-bit storage types only");
    constexpr int PIXB = 8;                         // 4 channels x 2 B
    constexpr int FRAME = STP_ROWS * STP_COLS * PIXB;   // 11840 B
    constexpr int NB = 2, KS = 2, FRAGB = 1024;
    constexpr int BTILE = NB * KS * FRAGB;          // 4 KiB per step
    constexpr int S = 49;
    constexpr int ITEMS = STP_ROWS * (STP_COLS / 4);   // 4-pixel items per frame
    constexpr int FQ = (ITEMS + 255) / 256;         // items per thread per frame (2)
    typedef u16x8 frag_t;

    __shared__ __attribute__((aligned(16))) unsigned char lds[3 * FRAME + 3 * BTILE];
    unsigned char* const ldsA = lds;
    unsigned char* const ldsB = lds + 3 * FRAME;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
#ifdef STEP_EMUL
    const int wave = tid >> 6;
#else
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
    const int khalf = lane >> 5;

    int t = blockIdx.x;
    const int tw_i = t % p.tiles_w; t /= p.tiles_w;
    const int th_i = t % p.tiles_h; t /= p.tiles_h;
    const int od = t % p.To;
    const int n = t / p.To;
    const int oh0 = th_i * 16, ow0 = tw_i * 16;
    const int nb0 = blockIdx.y * NB;

    const T* xg = (const T*)p.x;
    const unsigned char* wg = (const unsigned char*)p.w;
    const bool vec_ok = (p.W % 4) == 0;

    // ---- frame staging: LDS col cl <-> input col 2*ow0 - 4 + cl, row r <-> input row 2*oh0 - 2 + r
    struct Item { u16x4 c[3]; };
    auto load_frame = [&](int f, Item (&it)[FQ]) {
        const int ifr = 2 * od - 2 + f;
#pragma unroll
        for (int q = 0; q < FQ; ++q) {
            const int item = tid + q * 256;
            const int cq = item % (STP_COLS / 4), r = item / (STP_COLS / 4);
            const int ih = 2 * oh0 - 2 + r, iw0 = 2 * ow0 - 4 + cq * 4;
            const bool rowok = item < ITEMS && ifr >= 0 && ifr < p.T && ih >= 0 && ih < p.H;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                u16x4 v = {0, 0, 0, 0};
                if (rowok) {
                    const unsigned short* src = (const unsigned short*)xg + ((((size_t)n * p.T + ifr) * 3 + c) * p.H + ih) * p.W;
                    if (vec_ok && iw0 >= 0 && iw0 + 3 < p.W) {
                        v = *(const u16x4*)(src + iw0);
                    } else {
#pragma unroll
                        for (int a = 0; a < 4; ++a)
                            if (iw0 + a >= 0 && iw0 + a < p.W) v[a] = src[iw0 + a];
                    }
                }
                it[q].c[c] = v;
            }
        }
    };
    auto store_frame = [&](int slot, const Item (&it)[FQ]) {
#pragma unroll
        for (int q = 0; q < FQ; ++q) {
            const int item = tid + q * 256;
            if (item < ITEMS) {
                const int cq = item % (STP_COLS / 4), r = item / (STP_COLS / 4);
                unsigned char* dst = ldsA + slot * FRAME + (r * STP_COLS + cq * 4) * PIXB;
                const u16x8 lo = {it[q].c[0][0], it[q].c[1][0], it[q].c[2][0], 0, it[q].c[0][1], it[q].c[1][1], it[q].c[2][1], 0};
                const u16x8 hi = {it[q].c[0][2], it[q].c[1][2], it[q].c[2][2], 0, it[q].c[0][3], it[q].c[1][3], it[q].c[2][3], 0};
                *(u16x8*)dst = lo;
                *(u16x8*)(dst + 16) = hi;
            }
        }
    };

    // ---- weights: thread tid owns one 16-byte vector of the 4 KiB step tile
    const int bf = tid >> 6;                                   // fragment (nbl, j)
    const unsigned char* wthr = wg + (((size_t)min(nb0 + (bf >> 1), p.nblk32 - 1) * 98 + (bf & 1)) * 64 + (tid & 63)) * 16;
    auto load_B = [&](int s_) { return *(const u32x4*)(wthr + (size_t)s_ * 2 * FRAGB); };

    // ---- this lane's A base: pixel (2*th, 2*tw + 2) of the slot, + its k half
    const unsigned char* abase[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        const int th = wave * 4 + mb * 2 + ((lane & 31) >> 4), tw = lane & 15;
        abase[mb] = ldsA + ((2 * th) * STP_COLS + 2 * tw + 2 + 2 * khalf) * PIXB;
    }
    const unsigned char* const bwave = ldsB + lane * 16;

    f32x16 acc[2][NB];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][i][r] = 0.f;

    frag_t fa[2][KS][2], fb[2][KS][NB];
    auto read_frags = [&](auto setc, int bufoff, int aoff) {
        constexpr int SET = decltype(setc)::value;
#pragma unroll
        for (int j = 0; j < KS; ++j) {
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) fa[SET][j][mb] = *(const frag_t*)(abase[mb] + aoff + j * 32);
#pragma unroll
            for (int i = 0; i < NB; ++i) fb[SET][j][i] = *(const frag_t*)(bwave + bufoff + (i * KS + j) * FRAGB);
        }
    };
    auto mma_all = [&](auto setc) {
        constexpr int SET = decltype(setc)::value;
#pragma unroll
        for (int j = 0; j < KS; ++j)
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                mma_k16(fa[SET][j][0], fb[SET][j][i], acc[0][i], T());
                mma_k16(fa[SET][j][1], fb[SET][j][i], acc[1][i], T());
            }
    };

    // ---- prologue: frames 0 and 1, weight tiles 0 and 1, tile 2 in flight
    Item fr[FQ];
    load_frame(0, fr); store_frame(0, fr);
    load_frame(1, fr); store_frame(1, fr);
    u32x4 R = load_B(0);
    *(u32x4*)(ldsB + tid * 16) = R;
    R = load_B(1);
    *(u32x4*)(ldsB + BTILE + tid * 16) = R;
    R = load_B(2);
    __syncthreads();
    read_frags(std::integral_constant<int, 0>(), 0, 0);

    int b1 = BTILE, b2 = 2 * BTILE;
    int kd1 = 0, kh1 = 0;                              // coordinates of step s+1
    auto step = [&](auto setc, int s_) {
        constexpr int SET = decltype(setc)::value;
        const int kh = kh1, kd = kd1;                  // this step
        if (++kh1 == 7) { kh1 = 0; ++kd1; }
        if (kh == 0 && kd + 2 < 7) load_frame(kd + 2, fr);          // in flight over three steps
        if (s_ + 1 < S) read_frags(std::integral_constant<int, SET ^ 1>(), b1, (kd1 % 3) * FRAME + kh1 * (STP_COLS * PIXB));
        mma_all(setc);
        if (s_ + 2 < S) *(u32x4*)(ldsB + b2 + tid * 16) = R;
        if (s_ + 3 < S) R = load_B(s_ + 3);
        if (kh == 3 && kd + 2 < 7) store_frame((kd + 2) % 3, fr);   // the slot of frame kd-1 (last read 4+ steps ago)
        __syncthreads();
        const int nb = (b2 == 2 * BTILE) ? 0 : b2 + BTILE;
        b1 = b2; b2 = nb;
    };
#pragma unroll 1
    for (int s_ = 0; s_ < S; s_ += 2) {
        step(std::integral_constant<int, 0>(), s_);
        if (s_ + 1 < S) step(std::integral_constant<int, 1>(), s_ + 1);
    }

    // ---- epilogue: affine + ReLU, LDS transpose (fp32, 128 pixels at a time), 16-byte stores
    T* yg = (T*)p.y;
    constexpr int BN = NB * 32, G = BN / 8;
    float* ot = (float*)lds;
    float sc[NB], sh[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int co = min((nb0 + i) * 32 + (lane & 31), p.Cout - 1);
        sc[i] = p.scale ? p.scale[co] : 1.f;
        sh[i] = p.shift ? p.shift[co] : 0.f;
    }
    const bool vec_epi = (p.y_cstride % 8 == 0) && (p.y_coff % 8 == 0) && (p.Cout % 8 == 0) && (((uintptr_t)p.y) % 16 == 0);
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        if (mb) __syncthreads();
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                ot[(wave * 32 + cd_row(r, lane)) * BN + i * 32 + (lane & 31)] = fmaxf(acc[mb][i][r] * sc[i] + sh[i], 0.f);
        __syncthreads();
        for (int idx = tid; idx < 128 * G; idx += 256) {
            const int row = idx / G, g = idx % G;
            // row = wave*32 + rr ; pixel: th = wave*4 + mb*2 + (rr >> 4), tw = rr & 15
            const int oh = oh0 + (row >> 5) * 4 + mb * 2 + ((row & 31) >> 4), ow = ow0 + (row & 15);
            const int co = nb0 * 32 + g * 8;
            if (oh < p.Ho && ow < p.Wo && co < p.Cout) {
                const size_t opix = (((size_t)n * p.To + od) * p.Ho + oh) * p.Wo + ow;
                const float* src = ot + row * BN + g * 8;
                if (vec_epi) {
                    u16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = elem<T>::bits16(src[e]);
                    *(u16x8*)(yg + opix * p.y_cstride + p.y_coff + co) = o;
                } else {
                    for (int e = 0; e < 8; ++e)
                        if (co + e < p.Cout) yg[opix * p.y_cstride + p.y_coff + co + e] = elem<T>::from_f32(src[e]);
                }
            }
        }
    }
}

// --------------------------------------------------------------------------------------------
// stem_stream_kernel (16-bit types): the stem with a dense K axis.
// stem_tap_kernel keeps 4-channel pixels and 8 kw taps per row so that every fragment is one aligned
// 16-byte LDS read -- at the price of multiplying 32 K values per (kd, kh) row of which 21 are real.
// Here a frame row in LDS is the plain element stream [col][3 channels] (6 bytes per pixel, 240 bytes per
// row), so the 7 taps x 3 channels an output pixel needs from a row are 21 CONSECUTIVE elements starting at
// byte 12*tw + 12: three 8-element fragments (q = 0, 1, 2; the last 3 elements belong to the pixel after the
// window and meet zero weights).  Those addresses are only 4-byte aligned: a misaligned ds_read_b128
// measures 6.6x slower than an aligned one on gfx950, two ds_read2_b32 run at the full LDS rate
// (tools/ubench/lds_align.hip), so fragments are read as two 8-byte halves with 4-byte alignment.
// K order inside a frame (11 MFMA K-steps of 16, against 14 before):
//   j = 0..8 : rows (2*rp, 2*rp + 1), rp = j / 3, fragment q = j % 3; the lower lane half (k 0..7) takes
//              the even row, the upper half the odd row (+240 bytes);
//   j = 9    : row 6, q = 0 (lower half) and q = 1 (upper half);
//   j = 10   : row 6, q = 2 (lower half); the upper half multiplies zero weights.
// 77 K-steps per tile instead of 98.  Everything else follows stem_tap_kernel: 4 waves x (2 x 2 MFMA
// tiles) on a 16x16-pixel x 64-channel tile, 3-slot frame ring, weights through 3 LDS buffers (tiles of
// 4, 4 and 3 K-steps per frame: tile t of every frame lives in buffer t), one barrier per weight tile,
// fragments double-buffered in registers per K-step.
constexpr int STS_PITCH = 240;                      // bytes per LDS frame row (40 px x 3 ch x 2 B)
constexpr int STS_FRAME = STP_ROWS * STS_PITCH;     // 8880 B
constexpr int STS_KSTEPS = 77;

__device__ __forceinline__ u16x8 lds_read_frag_a4(const unsigned char* p) {
    typedef unsigned int u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));
    const u32x2_a4 lo = *(const u32x2_a4*)p;
    const u32x2_a4 hi = *(const u32x2_a4*)(p + 8);
    const u32x4 v = {lo.x, lo.y, hi.x, hi.y};
    u16x8 r;
    __builtin_memcpy(&r, &v, 16);
    return r;
}

template <typename T>
__global__ __launch_bounds__(256) STEP_WAVES_PER_SIMD(3) void stem_stream_kernel(StemParams p) {
    static_assert(sizeof(T) == 2, "16-bit storage types only");
    constexpr int FRAME = STS_FRAME, PITCH = STS_PITCH;
    constexpr int NB = 2, FRAGB = 1024;
    constexpr int NBREG = 4 * FRAGB;                // one n-block's share of a weight buffer (up to 4 K-steps)
    constexpr int BBUF = NB * NBREG;                // 8 KiB
    constexpr int ITEMS = STP_ROWS * (STP_COLS / 4);   // 4-pixel items per frame
    constexpr int FQ = (ITEMS + 255) / 256;
    constexpr int NTILES = 21;                      // weight tiles: 3 per frame
    typedef u16x8 frag_t;

    __shared__ __attribute__((aligned(16))) unsigned char lds[3 * FRAME + 3 * BBUF];
    unsigned char* const ldsA = lds;
    unsigned char* const ldsB = lds + 3 * FRAME;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
#ifdef STEP_EMUL
    const int wave = tid >> 6;
#else
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
    const int khalf = lane >> 5;

    int t = blockIdx.x;
    const int tw_i = t % p.tiles_w; t /= p.tiles_w;
    const int th_i = t % p.tiles_h; t /= p.tiles_h;
    const int od = t % p.To;
    const int n = t / p.To;
    const int oh0 = th_i * 16, ow0 = tw_i * 16;
    const int nb0 = blockIdx.y * NB;

    const T* xg = (const T*)p.x;
    const unsigned char* wg = (const unsigned char*)p.w;
    const bool vec_ok = (p.W % 4) == 0;

    // ---- frame staging: LDS col cl <-> input col 2*ow0 - 4 + cl, row r <-> input row 2*oh0 - 2 + r
    struct Item { u16x4 c[3]; };
    auto load_frame = [&](int f, Item (&it)[FQ]) {
        const int ifr = 2 * od - 2 + f;
#pragma unroll
        for (int q = 0; q < FQ; ++q) {
            const int item = tid + q * 256;
            const int cq = item % (STP_COLS / 4), r = item / (STP_COLS / 4);
            const int ih = 2 * oh0 - 2 + r, iw0 = 2 * ow0 - 4 + cq * 4;
            const bool rowok = item < ITEMS && ifr >= 0 && ifr < p.T && ih >= 0 && ih < p.H;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                u16x4 v = {0, 0, 0, 0};
                if (rowok) {
                    const unsigned short* src = (const unsigned short*)xg + ((((size_t)n * p.T + ifr) * 3 + c) * p.H + ih) * p.W;
                    if (vec_ok && iw0 >= 0 && iw0 + 3 < p.W) {
                        v = *(const u16x4*)(src + iw0);
                    } else {
#pragma unroll
                        for (int a = 0; a < 4; ++a)
                            if (iw0 + a >= 0 && iw0 + a < p.W) v[a] = src[iw0 + a];
                    }
                }
                it[q].c[c] = v;
            }
        }
    };
    auto store_frame = [&](int slotoff, const Item (&it)[FQ]) {
#pragma unroll
        for (int q = 0; q < FQ; ++q) {
            const int item = tid + q * 256;
            if (item < ITEMS) {
                const int cq = item % (STP_COLS / 4), r = item / (STP_COLS / 4);
                unsigned char* dst = ldsA + slotoff + r * PITCH + cq * 24;
                const u16x4 v0 = {it[q].c[0][0], it[q].c[1][0], it[q].c[2][0], it[q].c[0][1]};
                const u16x4 v1 = {it[q].c[1][1], it[q].c[2][1], it[q].c[0][2], it[q].c[1][2]};
                const u16x4 v2 = {it[q].c[2][2], it[q].c[0][3], it[q].c[1][3], it[q].c[2][3]};
                *(u16x4*)dst = v0;
                *(u16x4*)(dst + 8) = v1;
                *(u16x4*)(dst + 16) = v2;
            }
        }
    };

    // ---- weights: thread tid moves one 16-byte vector per n-block of a tile (K-steps kstep0 .. kstep0+3;
    //      the 3-K-step tiles carry one K-step of the next tile along, never read)
    const unsigned char* wthr[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i)
        wthr[i] = wg + ((size_t)min(nb0 + i, p.nblk32 - 1) * STS_KSTEPS) * FRAGB + tid * 16;
    struct BReg { u32x4 v[NB]; };
    auto load_B = [&](int tile) {
        tile = min(tile, NTILES - 1);
        const int ks0 = (tile / 3) * 11 + (tile % 3) * 4;
        BReg r;
#pragma unroll
        for (int i = 0; i < NB; ++i) r.v[i] = *(const u32x4*)(wthr[i] + (size_t)ks0 * FRAGB);
        return r;
    };
    auto store_B = [&](int buf, const BReg& r) {
#pragma unroll
        for (int i = 0; i < NB; ++i) *(u32x4*)(ldsB + buf * BBUF + i * NBREG + tid * 16) = r.v[i];
    };

    // ---- this lane's A base: element stream of row 2*th at pixel 2*tw + 2
    const unsigned char* abase[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        // rows {0, 2} of the wave's 4-row strip in block 0, {1, 3} in block 1: ds_read2_b32 is serviced per 32-lane
        // half with 32 banks, lanes 0-15 cover the banks 3*tw mod 32 and 4 input rows further down (960 B = 16 banks)
        // lanes 16-31 cover exactly the other 16 (rows {0, 1} together were a 2-way conflict on every fragment read)
        const int th = wave * 4 + mb + 2 * ((lane & 31) >> 4), tw = lane & 15;
        abase[mb] = ldsA + (2 * th) * PITCH + (2 * tw + 2) * 6;
    }
    const int kh_row = khalf * PITCH, kh_q = khalf * 16;
    const unsigned char* const bwave = ldsB + lane * 16;

    f32x16 acc[2][NB];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][i][r] = 0.f;

    frag_t fa[2][2], fb[2][NB];
    // fragments of K-step j (0..10) of the frame whose slot starts at byte `slotoff`
    auto read_frags = [&](auto setc, auto jc, int slotoff) {
        constexpr int SET = decltype(setc)::value;
        constexpr int J = decltype(jc)::value;
        const int aoff = slotoff + (J < 9 ? (2 * (J / 3)) * PITCH + (J % 3) * 16 + kh_row : (J == 9 ? 6 * PITCH + kh_q : 6 * PITCH + 32));
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) fa[SET][mb] = lds_read_frag_a4(abase[mb] + aoff);
#pragma unroll
        for (int i = 0; i < NB; ++i) fb[SET][i] = *(const frag_t*)(bwave + (J / 4) * BBUF + i * NBREG + (J % 4) * FRAGB);
    };
    auto mma_all = [&](auto setc) {
        constexpr int SET = decltype(setc)::value;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            mma_k16(fa[SET][0], fb[SET][i], acc[0][i], T());
            mma_k16(fa[SET][1], fb[SET][i], acc[1][i], T());
        }
    };

    // ---- prologue: frames 0 and 1, weight tiles 0 and 1, tile 2 in flight
    Item fr[FQ];
    load_frame(0, fr); store_frame(0, fr);
    load_frame(1, fr); store_frame(FRAME, fr);
    BReg R = load_B(0);
    store_B(0, R);
    R = load_B(1);
    store_B(1, R);
    R = load_B(2);
    __syncthreads();
    read_frags(std::integral_constant<int, 0>(), std::integral_constant<int, 0>(), 0);

    int cur = 0, nxt = FRAME, nn = 2 * FRAME;          // slot byte offsets of frames kd, kd+1, kd+2
    // one frame = 11 K-steps; P = parity of the frame's first K-step (11 is odd, so it alternates)
    auto frame_iter = [&](auto pc, int kd) {
        constexpr int P = decltype(pc)::value;
        if (kd + 2 < 7) load_frame(kd + 2, fr);                     // in flight until the second barrier
#define STS_KSTEP(J)                                                                                                  \
        {                                                                                                             \
            if (J < 10) read_frags(std::integral_constant<int, (P + J + 1) & 1>(), std::integral_constant<int, (J + 1) % 11>(), cur); \
            else        read_frags(std::integral_constant<int, (P + J + 1) & 1>(), std::integral_constant<int, 0>(), nxt);            \
            mma_all(std::integral_constant<int, (P + J) & 1>());                                                      \
        }
#define STS_TILE_END(TI)                                                                                              \
        {                                                                                                             \
            store_B((TI + 2) % 3, R);                  /* tile 3*kd + TI + 2 -> its home buffer */                  \
            R = load_B(3 * kd + TI + 3);                                                                              \
        }
        STS_KSTEP(0) STS_KSTEP(1) STS_KSTEP(2) STS_KSTEP(3)
        STS_TILE_END(0)
        __syncthreads();
        STS_KSTEP(4) STS_KSTEP(5) STS_KSTEP(6) STS_KSTEP(7)
        STS_TILE_END(1)
        if (kd + 2 < 7) store_frame(nn, fr);           // the slot frame kd-1 left (last read before the previous frame's last barrier)
        __syncthreads();
        STS_KSTEP(8) STS_KSTEP(9) STS_KSTEP(10)
        STS_TILE_END(2)
        __syncthreads();
#undef STS_KSTEP
#undef STS_TILE_END
        const int tmp = cur; cur = nxt; nxt = nn; nn = tmp;
    };
#pragma unroll 1
    for (int kd = 0; kd < 6; kd += 2) {
        frame_iter(std::integral_constant<int, 0>(), kd);
        frame_iter(std::integral_constant<int, 1>(), kd + 1);
    }
    frame_iter(std::integral_constant<int, 0>(), 6);

    // ---- epilogue: affine + ReLU, LDS transpose (fp32, 128 pixels at a time), 16-byte stores
    T* yg = (T*)p.y;
    constexpr int BN = NB * 32, G = BN / 8;
    float* ot = (float*)lds;
    float sc[NB], sh[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int co = min((nb0 + i) * 32 + (lane & 31), p.Cout - 1);
        sc[i] = p.scale ? p.scale[co] : 1.f;
        sh[i] = p.shift ? p.shift[co] : 0.f;
    }
    const bool vec_epi = (p.y_cstride % 8 == 0) && (p.y_coff % 8 == 0) && (p.Cout % 8 == 0) && (((uintptr_t)p.y) % 16 == 0);
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        if (mb) __syncthreads();
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                ot[(wave * 32 + cd_row(r, lane)) * BN + i * 32 + (lane & 31)] = fmaxf(acc[mb][i][r] * sc[i] + sh[i], 0.f);
        __syncthreads();
        for (int idx = tid; idx < 128 * G; idx += 256) {
            const int row = idx / G, g = idx % G;
            const int oh = oh0 + (row >> 5) * 4 + mb + 2 * ((row & 31) >> 4), ow = ow0 + (row & 15);
            const int co = nb0 * 32 + g * 8;
            if (oh < p.Ho && ow < p.Wo && co < p.Cout) {
                const size_t opix = (((size_t)n * p.To + od) * p.Ho + oh) * p.Wo + ow;
                const float* src = ot + row * BN + g * 8;
                if (vec_epi) {
                    u16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = elem<T>::bits16(src[e]);
                    *(u16x8*)(yg + opix * p.y_cstride + p.y_coff + co) = o;
                } else {
                    for (int e = 0; e < 8; ++e)
                        if (co + e < p.Cout) yg[opix * p.y_cstride + p.y_coff + co + e] = elem<T>::from_f32(src[e]);
                }
            }
        }
    }
}

// torch [Cout][3][7][7][7] fp32 -> [nb32][77 K-steps][lane][8] (+ one zero K-step at the very end) for stem_stream_kernel
template <typename T>
__global__ void stem_stream_pack_weight_kernel(const float* __restrict__ w, T* __restrict__ out, int Cout, long long total) {
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)blockDim.x * gridDim.x) {
        const int e = (int)(idx & 7);
        const int lane = (int)((idx >> 3) & 63);
        const long long q = idx >> 9;
        const int ks = (int)(q % STS_KSTEPS);
        const long long nb = q / STS_KSTEPS;
        const int kd = ks / 11, j = ks % 11, khalf = lane >> 5;
        int kh, fq;                                  // row and fragment of this lane half; fq < 0: zero
        if (j < 9) { kh = 2 * (j / 3) + khalf; fq = j % 3; }
        else if (j == 9) { kh = 6; fq = khalf; }
        else { kh = 6; fq = khalf ? -1 : 2; }
        const int se = 8 * fq + e, kw = se / 3, c = se % 3;
        const long long co = nb * 32 + (lane & 31);
        float v = 0.f;
        if (fq >= 0 && co < Cout && kw < 7) v = w[((((size_t)co * 3 + c) * 7 + kd) * 7 + kh) * 7 + kw];
        out[idx] = elem<T>::from_f32(v);
    }
}

// torch [Cout][3][7][7][7] fp32 -> [nb32][kd][kh][j][lane][8]; element e: kw = 4j + 2*(lane>>5) + (e>>2), c = e&3
template <typename T>
__global__ void stem_pack_weight_kernel(const float* __restrict__ w, T* __restrict__ out, int Cout, long long total) {
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)blockDim.x * gridDim.x) {
        const int e = (int)(idx & 7);
        const int lane = (int)((idx >> 3) & 63);
        long long q = idx >> 9;
        const int ks = (int)(q % 98);
        const int nb = (int)(q / 98);
        const int j = ks & 1, kh = (ks >> 1) % 7, kd = (ks >> 1) / 7;
        const int kw = 4 * j + 2 * (lane >> 5) + (e >> 2), c = e & 3;
        const int co = nb * 32 + (lane & 31);
        float v = 0.f;
        if (co < Cout && c < 3 && kw < 7) v = w[((((size_t)co * 3 + c) * 7 + kd) * 7 + kh) * 7 + kw];
        out[idx] = elem<T>::from_f32(v);
    }
}

// ---- host-side dispatch --------------------------------------------------------------------
static inline unsigned flat_grid(long long total, int block) {
    long long g = ceil_div64(total, block);
    if (g > 16384) g = 16384;
    return (unsigned)g;
}

template <typename T, int TWL, int KD, int KH, int KW, bool FLAT, int CKT>
static int launch_nb(const ConvParams& p, int NB, dim3 grid, step_stream_t stream) {
    switch (NB) {
        case 1: STEP_LAUNCH((conv_igemm_kernel<T, TWL, 1, KD, KH, KW, FLAT, CKT>), grid, dim3(256), stream, p); break;
        case 2: STEP_LAUNCH((conv_igemm_kernel<T, TWL, 2, KD, KH, KW, FLAT, CKT>), grid, dim3(256), stream, p); break;
        case 3: STEP_LAUNCH((conv_igemm_kernel<T, TWL, 3, KD, KH, KW, FLAT, CKT>), grid, dim3(256), stream, p); break;
        default: STEP_LAUNCH((conv_igemm_kernel<T, TWL, 4, KD, KH, KW, FLAT, CKT>), grid, dim3(256), stream, p); break;
    }
    return STEP_LAUNCH_CHECK();
}

// pick the accumulator depth: least padded work first, then the deepest tile that still gives the
// chip enough workgroups (256 CUs, several resident workgroups each)
static int pick_nb(int nblk32, long long mtiles) {
    int best = 1;
    long long best_cost = -1;
    const int cand[4] = {4, 3, 2, 1};
    for (int k = 0; k < 4; ++k) {
        const int nb = cand[k];
        const long long groups = ceil_div(nblk32, nb);
        const long long wgs = groups * mtiles;
        // time ~ (#rounds of 1024 resident workgroups) x (per-workgroup time: nb MFMA units + ~2 units of
        // slab staging); ties go to the deeper tile (less re-staging traffic)
        const long long cost = ceil_div64(wgs, 1024) * (nb + 2) * 8 - nb;
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = nb; }
    }
    return best;
}

// Which instantiation a descriptor maps to (also used by step_conv_kernel_name so that bench.py can
// attribute time and work to the kernel name rocprofv3 reports).
//   impl 0: conv_igemm_kernel (4 waves, 128-px tile, weights straight from L2)  -- 1x1x1 and small problems
//   impl 1: conv_tap_kernel   (8 waves, 256-px tile, weights through an LDS-DMA double buffer)
struct ConvPlan { bool ok, flat, wide, deep; int impl, NB, tps, mb, twl, tiles_h, tiles_w, tiles_d, gtd, gth, gtw, ksplit, kchunk16, mbk, mpad, cpad; long long mtiles; };

static int pick_nb_tap(int nblk32, long long mtiles) {
    static const int forced = getenv("STEP_CONV_NB") ? atoi(getenv("STEP_CONV_NB")) : 0;     // tuning aid
    if (forced >= 1 && forced <= 3) return forced;
    int best = 1;
    double best_cost = -1;
    for (int nb = 3; nb >= 1; --nb) {   // 2 fragment sets + 2*nb accumulators must fit 256 VGPRs: nb <= 3
        const long long groups = ceil_div(nblk32, 2 * nb);
        const long long wgs = groups * mtiles;
        // one 512-thread workgroup per CU: rounds of 256; per-workgroup time ~ 2*nb MFMA units + fixed part
        const double cost = (double)ceil_div64(wgs, 256) * (2.0 * nb + 1.5) - 0.01 * nb;
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = nb; }
    }
    return best;
}

static int conv_impl_override() {   // tuning aid: STEP_CONV_IMPL=igemm|tap|tap2 forces one implementation
    const char* e = getenv("STEP_CONV_IMPL");
    if (!e) return -1;
    if (e[0] == 'i') return 0;
    if (e[0] == 'p') return 5;      // pw: force the streaming pointwise GEMM for every 1x1x1 conv
    if (e[0] == 't') return (e[1] && e[2] && e[3] == '2') ? 2 : ((e[1] && e[2] && e[3] == '4') ? 4 : 1);
    return -1;
}

// A kernel with kd == 1 never looks across planes, so clips and frames are one axis: fold N into D.  The
// plane-folded tile shape then also applies to [N*T, 1, 7, 7, C] head tensors without the caller reshaping.
static step_conv_desc canonical_desc(const step_conv_desc* d) {
    step_conv_desc e = *d;
    if (e.kd == 1 && (long long)e.N * e.D <= 0x7fffffffLL) { e.D *= e.N; e.N = 1; }
    return e;
}

// best general box (td, th, tw) for conv_tap_kernel<TWL = 0>: td*th*tw <= 256 pixels, halo within the LDS
// reservation; fewest tiles wins, then the smaller halo.  Rows stay wide (the whole map width or half of it):
// a box of short rows, e.g. 16x4x4 on a 28x28 map, needs the fewest tiles (49 against 64) but measured 70 % more
// time per tile -- eight 4-pixel rows per MFMA row block conflict in LDS and the halo is 2.5x the tile.
static long long best_gen_box(int D, int H, int W, int kd, int npix_limit, int* btd, int* bth, int* btw) {
    long long best = -1; int bhalo = 0;
    for (int kw_ = 1; kw_ <= 8; ++kw_) {
        const int tw = ceil_div(W, kw_);
        if (tw > 256) continue;
        if (kw_ > 2 && tw < 16) break;
        for (int kh_ = 1; kh_ <= H; ++kh_) {
            const int th = ceil_div(H, kh_);
            if (th * tw > 256) continue;
            if (kh_ > 1 && th == ceil_div(H, kh_ - 1)) continue;
            for (int td = 1; td <= D && td * th * tw <= 256; ++td) {
                const int halo = (td + kd - 1) * (th + 2) * (tw + 2);
                if (halo > npix_limit) break;
                const long long tiles = (long long)ceil_div(D, td) * ceil_div(H, th) * ceil_div(W, tw);
                if (best < 0 || tiles < best || (tiles == best && halo < bhalo)) { best = tiles; bhalo = halo; *btd = td; *bth = th; *btw = tw; }
            }
        }
    }
    return best;
}

static ConvPlan conv_plan(const step_conv_desc* d) {
    ConvPlan pl;
    pl.ok = true; pl.wide = false; pl.tiles_h = pl.tiles_w = 0; pl.impl = 0; pl.tps = 1; pl.mb = 2; pl.deep = false; pl.twl = 4; pl.tiles_d = d->D; pl.gtd = pl.gth = pl.gtw = 1; pl.ksplit = 0; pl.kchunk16 = 0; pl.mbk = 0; pl.mpad = 0; pl.cpad = 0;
    const int nblk32 = ceil_div(d->Cout, 32);
    const bool k1 = d->kd == 1 && d->kh == 1 && d->kw == 1;
    const bool k333 = d->kd == 3 && d->kh == 3 && d->kw == 3;
    const bool k133 = d->kd == 1 && d->kh == 3 && d->kw == 3;
    pl.flat = k1;
    if (k1) {
        pl.mtiles = ceil_div64((long long)d->N * d->D * d->H * d->W, 128);
        pl.NB = pick_nb(nblk32, pl.mtiles);
        // deep-K pointwise convs on enough pixels: the streaming 8-wave GEMM (STEP_CONV_IMPL=igemm forces the other)
        const long long mt256 = ceil_div64((long long)d->N * d->D * d->H * d->W, 256);
        const int ov1 = conv_impl_override();
        if (ov1 == 5 || (ov1 != 0 && d->Cin >= 128 && d->Cout >= 64 && mt256 * ceil_div(nblk32, 2) >= 32)) {
            pl.impl = 2;
            pl.mtiles = mt256;
            pl.NB = pick_nb_tap(nblk32, mt256);
            return pl;
        }
        // few rows x very deep K (the heads' Linear layers): split K over the chip (needs the workspace of
        // step_conv_forward_ws; STEP_CONV_SPLITK=0 disables)
        static const bool splitk_ok = !(getenv("STEP_CONV_SPLITK") && atoi(getenv("STEP_CONV_SPLITK")) == 0);
        const long long M = (long long)d->N * d->D * d->H * d->W;
        if (splitk_ok && ov1 != 0 && d->dtype != STEP_F32 && d->Cin >= 2048 && M <= 1024 && pl.mtiles * nblk32 < 64) {
            pl.impl = 3;
            pl.mbk = M <= 32 ? 1 : (M <= 64 ? 2 : 4);
            const int KC16 = ceil_div(d->Cin, CK) * 2;
            const long long mt = ceil_div64(M, pl.mbk * 32);
            // ~512 workgroups, at least 8 k16 steps (2 per wave) each
            int ks = (int)(512 / (mt * nblk32));
            if (ks < 1) ks = 1;
            pl.kchunk16 = ceil_div(KC16, ks);
            if (pl.kchunk16 < 8) pl.kchunk16 = 8;
            pl.ksplit = ceil_div(KC16, pl.kchunk16);
            pl.mpad = (int)(mt * pl.mbk * 32);
            pl.cpad = nblk32 * 32;
            pl.mtiles = mt;
            pl.NB = 1;
            return pl;
        }
        pl.deep = d->Cin >= 256;        // 128-channel slabs: 4x fewer barriers along a deep K
        if (const char* e = getenv("STEP_CONV_DEEP")) pl.deep = (e[0] == '1');   // tuning aid
        return pl;
    }
    if (!k333 && !k133) { pl.ok = false; pl.mtiles = 0; pl.NB = 1; return pl; }
    const int ov = conv_impl_override();
    // 256-pixel tiles: 1 plane x 16x16, 1 x 8x32 or 4 planes x 8x8 -- whichever covers N x D x H x W with the
    // fewest tiles (ties: in that order)
    const long long t16 = (long long)d->D * ceil_div(d->H, 16) * ceil_div(d->W, 16);
    const long long t32 = (long long)d->D * ceil_div(d->H, 8) * ceil_div(d->W, 32);
    const long long t8 = (long long)ceil_div(d->D, 4) * ceil_div(d->H, 8) * ceil_div(d->W, 8);
    int twl = 4;
    long long tbest = t16;   // (per clip)
    if (t32 < tbest) { tbest = t32; twl = 5; }
    if (t8 < tbest) { tbest = t8; twl = 3; }
    // a general box when it needs at least 7 % fewer tiles than the best power-of-two shape (its staging index
    // arithmetic divides and its LDS reads are not conflict-free, ~5 % per tile).  Measured: C2 (28x28 / 14x14 maps,
    // 12.5 % fewer tiles) +1.2 % clips/s, the 400x400 backbone (50x50 / 25x25 / 100x100 maps, 28 % fewer tiles) +9 %.
    // STEP_CONV_GEN=<percent> moves the threshold, 0 disables the general boxes.
    int gtd = 1, gth = 1, gtw = 1;
    static const int gen_pct = getenv("STEP_CONV_GEN") ? atoi(getenv("STEP_CONV_GEN")) : 93;    // 0 disables
    const int twl_p2 = twl;
    const long long tbest_p2 = tbest;
    if (gen_pct > 0) {
        const long long tg = best_gen_box(d->D, d->H, d->W, d->kd, CONV_GEN_NPIX, &gtd, &gth, &gtw);
        if (tg > 0 && tg * 100 <= tbest * gen_pct) { tbest = tg; twl = 0; }
        if (twl == 0 && pick_nb_tap(nblk32, (long long)d->N * tbest) == 1) {
            // the NB = 1 instantiation reserves a smaller halo (two workgroups per CU): the box must fit it
            int std_ = 1, sth = 1, stw = 1;
            const long long ts = best_gen_box(d->D, d->H, d->W, d->kd, CONV_GEN_NPIX_SMALL, &std_, &sth, &stw);
            if (ts > 0 && ts * 100 <= tbest_p2 * gen_pct && pick_nb_tap(nblk32, (long long)d->N * ts) == 1) {
                tbest = ts; gtd = std_; gth = sth; gtw = stw;
            } else {
                tbest = tbest_p2; twl = twl_p2;
            }
        }
    }
    const long long mt256 = (long long)d->N * tbest;
    // few-tile, small-Cin problems stay on the 4-wave 128-pixel kernel (more workgroups); everything else -- the
    // Cin = 16/32 branches of the Inception blocks included (measured: 9 x 3x3x3 layers 0.253 ms against 0.317 ms) --
    // runs the pipelined kernel
    const bool use_tap = ov >= 1 || (ov != 0 && (d->Cin >= 64 || mt256 >= 32));
    if (use_tap) {
        pl.impl = 1;
        pl.tps = (ov == 1) ? 1 : 2;     // two taps per barrier measured 6-15 % faster than one (STEP_CONV_IMPL=tap forces one)
        pl.mb = 2;                      // (the 4-wave MB = 4 form of the kernel template spills at NB >= 2 and is not instantiated)
        pl.twl = twl;
        pl.wide = twl == 5;
        pl.gtd = gtd; pl.gth = gth; pl.gtw = gtw;
        if (twl == 0) {
            pl.tiles_d = ceil_div(d->D, gtd); pl.tiles_h = ceil_div(d->H, gth); pl.tiles_w = ceil_div(d->W, gtw);
        } else {
            pl.tiles_d = twl == 3 ? ceil_div(d->D, 4) : d->D;
            pl.tiles_h = twl == 4 ? ceil_div(d->H, 16) : ceil_div(d->H, 8);
            pl.tiles_w = twl == 4 ? ceil_div(d->W, 16) : (twl == 5 ? ceil_div(d->W, 32) : ceil_div(d->W, 8));
        }
        pl.mtiles = mt256;
        pl.NB = pick_nb_tap(nblk32, pl.mtiles);
        return pl;
    }
    // 128-pixel tiles: 8x16 or 4x32
    const long long w16 = (long long)ceil_div(d->H, 8) * ceil_div(d->W, 16);
    const long long w32 = (long long)ceil_div(d->H, 4) * ceil_div(d->W, 32);
    pl.wide = w32 < w16;
    pl.tiles_h = pl.wide ? ceil_div(d->H, 4) : ceil_div(d->H, 8);
    pl.tiles_w = pl.wide ? ceil_div(d->W, 32) : ceil_div(d->W, 16);
    pl.mtiles = (long long)d->N * d->D * pl.tiles_h * pl.tiles_w;
    pl.NB = pick_nb(nblk32, pl.mtiles);
    return pl;
}

template <typename T, int TWL, int KD, int KH, int KW>
static int launch_tap(const ConvParams& p, int NB, int tps, int mb, dim3 grid, step_stream_t stream) {
#define STEP_TAP(NB_, TPS_, MB_) STEP_LAUNCH((conv_tap_kernel<T, TWL, NB_, KD, KH, KW, TPS_, MB_>), grid, dim3(MB_ == 2 ? 512 : 256), stream, p)
    if (tps == 2) {
        switch (NB) {
            case 1: STEP_TAP(1, 2, 2); break;
            case 2: STEP_TAP(2, 2, 2); break;
            default: STEP_TAP(3, 2, 2); break;
        }
    } else {
        switch (NB) {
            case 1: STEP_TAP(1, 1, 2); break;
            case 2: STEP_TAP(2, 1, 2); break;
            default: STEP_TAP(3, 1, 2); break;
        }
    }
#undef STEP_TAP
    return STEP_LAUNCH_CHECK();
}

template <typename T>
static int splitk_forward_t(const ConvPlan& pl, const ConvParams& p, float* ws, step_stream_t stream) { return STEP_E_UNSUPPORTED; }
template <typename T16>
static int splitk_forward_16(const ConvPlan& pl, const ConvParams& p, float* ws, step_stream_t stream) {
    dim3 grid((unsigned)p.nblk32, (unsigned)pl.ksplit, (unsigned)pl.mtiles);
    switch (pl.mbk) {
        case 1: STEP_LAUNCH((pw_splitk_kernel<T16, 1>), grid, dim3(256), stream, p, ws, pl.kchunk16, pl.mpad, pl.cpad); break;
        case 2: STEP_LAUNCH((pw_splitk_kernel<T16, 2>), grid, dim3(256), stream, p, ws, pl.kchunk16, pl.mpad, pl.cpad); break;
        default: STEP_LAUNCH((pw_splitk_kernel<T16, 4>), grid, dim3(256), stream, p, ws, pl.kchunk16, pl.mpad, pl.cpad); break;
    }
    const long long total = p.Mtot * p.Cout;
    STEP_LAUNCH((pw_splitk_finish_kernel<T16>), dim3(flat_grid(total, 256)), dim3(256), stream, p, (const float*)ws, pl.ksplit, pl.mpad, pl.cpad);
    return STEP_LAUNCH_CHECK();
}
template <> int splitk_forward_t<bf16_t>(const ConvPlan& pl, const ConvParams& p, float* ws, step_stream_t stream) { return splitk_forward_16<bf16_t>(pl, p, ws, stream); }
template <> int splitk_forward_t<f16_t>(const ConvPlan& pl, const ConvParams& p, float* ws, step_stream_t stream) { return splitk_forward_16<f16_t>(pl, p, ws, stream); }

template <typename T>
static int conv_forward_t(const step_conv_desc* d, ConvParams p, void* ws, size_t ws_bytes, step_stream_t stream) {
    constexpr int VEC = elem<T>::VEC;
    if (d->Cin % VEC || d->x_cstride % VEC || d->x_coff % VEC) return STEP_E_ALIGN;
    if (((uintptr_t)p.x % 16) || ((uintptr_t)p.w % 16)) return STEP_E_ALIGN;
    ConvPlan pl = conv_plan(d);
    if (!pl.ok) return STEP_E_UNSUPPORTED;
    if (pl.impl == 3) {
        const size_t need = (size_t)pl.ksplit * pl.mpad * pl.cpad * sizeof(float);
        if (ws && ws_bytes >= need && ((uintptr_t)ws % 16) == 0) return splitk_forward_t<T>(pl, p, (float*)ws, stream);
        // no workspace (plain step_conv_forward): the tiled kernel
        pl.impl = 0; pl.mtiles = ceil_div64(p.Mtot, 128); pl.NB = pick_nb(p.nblk32, pl.mtiles); pl.deep = d->Cin >= 256;
    }
    p.tiles_h = pl.tiles_h; p.tiles_w = pl.tiles_w; p.tiles_d = pl.tiles_d;
    p.gtd = pl.gtd; p.gth = pl.gth; p.gtw = pl.gtw;
    auto grid1d = [&](int groups) {                   // logical (mtiles x groups) grid as a 1-D launch padded to 8
        p.gx = (int)pl.mtiles; p.gy = groups;
        const long long tot = pl.mtiles * groups;
        return dim3((unsigned)((tot + 7) / 8 * 8));
    };
    if (pl.impl == 2) {
        dim3 grid = grid1d(ceil_div(p.nblk32, 2 * pl.NB));
        switch (pl.NB) {
            case 1: STEP_LAUNCH((conv_pw_kernel<T, 1>), grid, dim3(512), stream, p); break;
            case 2: STEP_LAUNCH((conv_pw_kernel<T, 2>), grid, dim3(512), stream, p); break;
            default: STEP_LAUNCH((conv_pw_kernel<T, 3>), grid, dim3(512), stream, p); break;
        }
        return STEP_LAUNCH_CHECK();
    }
    if (pl.impl == 1) {
        dim3 grid = grid1d(ceil_div(p.nblk32, 2 * pl.NB));
        if (d->kd == 3) {
            if (pl.twl == 0) return launch_tap<T, 0, 3, 3, 3>(p, pl.NB, pl.tps, pl.mb, grid, stream);
            if (pl.twl == 3) return launch_tap<T, 3, 3, 3, 3>(p, pl.NB, pl.tps, pl.mb, grid, stream);
            return pl.wide ? launch_tap<T, 5, 3, 3, 3>(p, pl.NB, pl.tps, pl.mb, grid, stream) : launch_tap<T, 4, 3, 3, 3>(p, pl.NB, pl.tps, pl.mb, grid, stream);
        }
        if (pl.twl == 0) return launch_tap<T, 0, 1, 3, 3>(p, pl.NB, pl.tps, pl.mb, grid, stream);
        if (pl.twl == 3) return launch_tap<T, 3, 1, 3, 3>(p, pl.NB, pl.tps, pl.mb, grid, stream);
        return pl.wide ? launch_tap<T, 5, 1, 3, 3>(p, pl.NB, pl.tps, pl.mb, grid, stream) : launch_tap<T, 4, 1, 3, 3>(p, pl.NB, pl.tps, pl.mb, grid, stream);
    }
    dim3 grid = grid1d(ceil_div(p.nblk32, pl.NB));
    if (pl.flat)
        return pl.deep ? launch_nb<T, 4, 1, 1, 1, true, 128>(p, pl.NB, grid, stream) : launch_nb<T, 4, 1, 1, 1, true, 32>(p, pl.NB, grid, stream);
    if (d->kd == 3)
        return pl.wide ? launch_nb<T, 5, 3, 3, 3, false, 32>(p, pl.NB, grid, stream) : launch_nb<T, 4, 3, 3, 3, false, 32>(p, pl.NB, grid, stream);
    return pl.wide ? launch_nb<T, 5, 1, 3, 3, false, 32>(p, pl.NB, grid, stream) : launch_nb<T, 4, 1, 3, 3, false, 32>(p, pl.NB, grid, stream);
}

template <typename T>
static int stem_forward_t(StemParams p, step_stream_t stream) {
    dim3 grid((unsigned)((long long)p.N * p.To * p.tiles_h * p.tiles_w), (unsigned)ceil_div(p.nblk32, 2));
    STEP_LAUNCH((stem_igemm_kernel<T, 2>), grid, dim3(256), stream, p);
    return STEP_LAUNCH_CHECK();
}

static inline size_t stem_stream_offset(int Cout) { return (size_t)ceil_div(Cout, 32) * 98 * 512; }

template <typename T>
static int stem_stream_forward_t(StemParams p, step_stream_t stream) {
    p.w = (const T*)p.w + stem_stream_offset(p.Cout);
    p.tiles_h = ceil_div(p.Ho, 16); p.tiles_w = ceil_div(p.Wo, 16);
    dim3 grid((unsigned)((long long)p.N * p.To * p.tiles_h * p.tiles_w), (unsigned)ceil_div(p.nblk32, 2));
    STEP_LAUNCH((stem_stream_kernel<T>), grid, dim3(256), stream, p);
    return STEP_LAUNCH_CHECK();
}

template <typename T>
static int stem_tap_forward_t(StemParams p, step_stream_t stream) {
    p.tiles_h = ceil_div(p.Ho, 16); p.tiles_w = ceil_div(p.Wo, 16);
    dim3 grid((unsigned)((long long)p.N * p.To * p.tiles_h * p.tiles_w), (unsigned)ceil_div(p.nblk32, 2));
    STEP_LAUNCH((stem_tap_kernel<T>), grid, dim3(256), stream, p);
    return STEP_LAUNCH_CHECK();
}

}  // namespace step

using namespace step;

extern "C" {

size_t step_conv_packed_elems(int Cout, int Cin, int kd, int kh, int kw) {
    return (size_t)ceil_div(Cout, 32) * taps_padded(kd * kh * kw) * (ceil_div(Cin, CK) * 2) * 512;
}

int step_conv_pack_weight(const float* w, int Cout, int Cin, int kd, int kh, int kw, int dtype, const int32_t* perm,
                          void* packed, step_stream_t stream) {
    if (Cout <= 0 || Cin <= 0 || kd <= 0 || kh <= 0 || kw <= 0) return STEP_E_SHAPE;
    if (!w || !packed) return STEP_E_NULL;
    const long long total = (long long)step_conv_packed_elems(Cout, Cin, kd, kh, kw);
    const int KC16 = ceil_div(Cin, CK) * 2, ntaps = kd * kh * kw;
    const dim3 grid(flat_grid(total, 256));
    switch (dtype) {
        case STEP_F32: STEP_LAUNCH((pack_weight_kernel<float>), grid, dim3(256), stream, w, perm, (float*)packed, Cout, Cin, ntaps, KC16, total); break;
        case STEP_BF16: STEP_LAUNCH((pack_weight_kernel<bf16_t>), grid, dim3(256), stream, w, perm, (bf16_t*)packed, Cout, Cin, ntaps, KC16, total); break;
        case STEP_F16: STEP_LAUNCH((pack_weight_kernel<f16_t>), grid, dim3(256), stream, w, perm, (f16_t*)packed, Cout, Cin, ntaps, KC16, total); break;
        default: return STEP_E_DTYPE;
    }
    return STEP_LAUNCH_CHECK();
}

int step_conv_wgrad(const step_conv_desc* d, const void* x, const float* dy, float* dw, int accumulate, step_stream_t stream) {
    if (!d) return STEP_E_NULL;
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
        if (ch < 64) ch = 64;
        if (ch > 65536) ch = 65536;
        const int chunk = (int)ch;
        // (n, d, h) collapse into full chunks; the ragged tail is a second launch
        const long long full = M / chunk;
        const int tail = (int)(M % chunk);
        int rc = STEP_OK;
        auto launch = [&](long long jobs, int W, size_t pix0) {
            p.N = 1; p.D = (int)jobs; p.H = 1; p.W = W; p.jobs = jobs; p.rows = 1; p.total_rows = jobs;
            p.x = (const char*)x + pix0 * d->x_cstride * (d->dtype == STEP_F32 ? 4 : 2);
            p.dy = dy + pix0 * d->y_cstride;
            const bool narrow = d->Cin <= 32;
            p.cot = ceil_div(d->Cout, 64); p.cit = ceil_div(d->Cin, narrow ? 32 : 64);
            dim3 grid((unsigned)ceil_div64(jobs, 4), (unsigned)(p.cot * p.cit));
#define STEP_WG(T_) do { if (narrow) STEP_LAUNCH((conv_wgrad_kernel<T_, 2, 1>), grid, dim3(256), stream, p); \
                         else STEP_LAUNCH((conv_wgrad_kernel<T_, 2, 2>), grid, dim3(256), stream, p); } while (0)
            switch (d->dtype) {
                case STEP_F32: STEP_WG(float); break;
                case STEP_BF16: STEP_WG(bf16_t); break;
                case STEP_F16: STEP_WG(f16_t); break;
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
        if (rows < 1) rows = 1;
        if (rows > 0x3fffffff) rows = 0x3fffffff;
        p.rows = (int)rows;
    }
    p.jobs = ceil_div64(p.total_rows, p.rows);
    if (gy > 65535) return STEP_E_UNSUPPORTED;
    dim3 grid((unsigned)ceil_div64(p.jobs, 4), (unsigned)gy);
    switch (d->dtype) {
        case STEP_F32: STEP_WG(float); break;
        case STEP_BF16: STEP_WG(bf16_t); break;
        case STEP_F16: STEP_WG(f16_t); break;
        default: return STEP_E_DTYPE;
    }
#undef STEP_WG
    return STEP_LAUNCH_CHECK();
}

size_t step_conv_workspace_bytes(const step_conv_desc* d) {
    if (!d || d->N <= 0 || d->D <= 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->Cout <= 0) return 0;
    const step_conv_desc canon = canonical_desc(d);
    const ConvPlan pl = conv_plan(&canon);
    return (pl.ok && pl.impl == 3) ? (size_t)pl.ksplit * pl.mpad * pl.cpad * sizeof(float) : 0;
}

int step_conv_forward(const step_conv_desc* d, const void* x, const void* w_packed, const float* scale,
                      const float* shift, const void* res, void* y, void* y2, step_stream_t stream) {
    return step_conv_forward_ws(d, x, w_packed, scale, shift, res, y, y2, nullptr, 0, stream);
}

int step_conv_forward_ws(const step_conv_desc* d, const void* x, const void* w_packed, const float* scale,
                         const float* shift, const void* res, void* y, void* y2, void* ws, size_t ws_bytes, step_stream_t stream) {
    if (!d) return STEP_E_NULL;
    if (d->N < 0 || d->D <= 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->Cout <= 0) return STEP_E_SHAPE;
    const int split = (d->split > 0 && d->split < d->Cout) ? d->split : 0;
    const int cout_y = split ? split : d->Cout;
    if (d->x_coff < 0 || d->x_coff + d->Cin > d->x_cstride || d->y_coff < 0 || d->y_coff + cout_y > d->y_cstride)
        return STEP_E_SHAPE;
    if (split) {
        if (!(d->kd == 1 && d->kh == 1 && d->kw == 1)) return STEP_E_UNSUPPORTED;
        if (!y2) return STEP_E_NULL;
        if (d->y2_coff < 0 || d->y2_coff + (d->Cout - split) > d->y2_cstride) return STEP_E_SHAPE;
    }
    if (res && (d->res_coff < 0 || d->res_coff + d->Cout > d->res_cstride)) return STEP_E_SHAPE;
    if (d->N == 0) return STEP_OK;
    if (!x || !w_packed || !y) return STEP_E_NULL;
    const step_conv_desc canon = canonical_desc(d);
    d = &canon;
    ConvParams p;
    p.x = x; p.w = w_packed; p.scale = scale; p.shift = shift; p.res = res; p.y = y; p.y2 = y2;
    p.split = split; p.y2_cstride = d->y2_cstride; p.y2_coff = d->y2_coff;
    p.N = d->N; p.D = d->D; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Cout = d->Cout;
    p.x_cstride = d->x_cstride; p.x_coff = d->x_coff; p.y_cstride = d->y_cstride; p.y_coff = d->y_coff;
    p.r_cstride = d->res_cstride; p.r_coff = d->res_coff;
    p.relu = d->relu;
    p.tiles_h = p.tiles_w = 0; p.tiles_d = d->D; p.gtd = p.gth = p.gtw = 1; p.gx = p.gy = 0;
    p.nchunks = ceil_div(d->Cin, CK);
    p.nchunks32 = p.nchunks;
    p.vec_epi = (d->y_cstride % 8 == 0) && (d->y_coff % 8 == 0) && (d->Cout % 8 == 0) && (((uintptr_t)y) % 16 == 0) &&
                (!res || ((d->res_cstride % 8 == 0) && (d->res_coff % 8 == 0) && (((uintptr_t)res) % 16 == 0))) &&
                (!split || ((split % 8 == 0) && (d->y2_cstride % 8 == 0) && (d->y2_coff % 8 == 0) && (((uintptr_t)y2) % 16 == 0)));
    p.nblk32 = ceil_div(d->Cout, 32);
    p.Mtot = (long long)d->N * d->D * d->H * d->W;
    switch (d->dtype) {
        case STEP_F32: return conv_forward_t<float>(d, p, ws, ws_bytes, stream);
        case STEP_BF16: return conv_forward_t<bf16_t>(d, p, ws, ws_bytes, stream);
        case STEP_F16: return conv_forward_t<f16_t>(d, p, ws, ws_bytes, stream);
    }
    return STEP_E_DTYPE;
}

// two images back to back: [nb32][98 K-steps] (stem_igemm_kernel / stem_tap_kernel) and
// [nb32][77 K-steps] + one zero K-step (stem_stream_kernel, 16-bit types)
size_t step_stem_packed_elems(int Cout) { return stem_stream_offset(Cout) + ((size_t)ceil_div(Cout, 32) * STS_KSTEPS + 1) * 512; }

int step_stem_pack_weight(const float* w, int Cout, int dtype, void* packed, step_stream_t stream) {
    if (Cout <= 0) return STEP_E_SHAPE;
    if (!w || !packed) return STEP_E_NULL;
    const long long total = (long long)stem_stream_offset(Cout);
    const long long total2 = (long long)step_stem_packed_elems(Cout) - total;
    const dim3 grid(flat_grid(total, 256)), grid2(flat_grid(total2, 256));
    switch (dtype) {
        case STEP_F32:
            STEP_LAUNCH((stem_pack_weight_kernel<float>), grid, dim3(256), stream, w, (float*)packed, Cout, total);
            STEP_LAUNCH((stem_stream_pack_weight_kernel<float>), grid2, dim3(256), stream, w, (float*)packed + total, Cout, total2);
            break;
        case STEP_BF16:
            STEP_LAUNCH((stem_pack_weight_kernel<bf16_t>), grid, dim3(256), stream, w, (bf16_t*)packed, Cout, total);
            STEP_LAUNCH((stem_stream_pack_weight_kernel<bf16_t>), grid2, dim3(256), stream, w, (bf16_t*)packed + total, Cout, total2);
            break;
        case STEP_F16:
            STEP_LAUNCH((stem_pack_weight_kernel<f16_t>), grid, dim3(256), stream, w, (f16_t*)packed, Cout, total);
            STEP_LAUNCH((stem_stream_pack_weight_kernel<f16_t>), grid2, dim3(256), stream, w, (f16_t*)packed + total, Cout, total2);
            break;
        default: return STEP_E_DTYPE;
    }
    return STEP_LAUNCH_CHECK();
}

int step_stem_forward(int dtype, const void* x, int N, int T, int H, int W, const void* w_packed, const float* scale,
                      const float* shift, int Cout, void* y, int y_cstride, int y_coff, step_stream_t stream) {
    if (N < 0 || T <= 0 || H <= 0 || W <= 0 || Cout <= 0) return STEP_E_SHAPE;
    if (y_coff < 0 || y_coff + Cout > y_cstride) return STEP_E_SHAPE;
    if (N == 0) return STEP_OK;
    if (!x || !w_packed || !y) return STEP_E_NULL;
    if (((uintptr_t)x % 16) || ((uintptr_t)w_packed % 16)) return STEP_E_ALIGN;
    StemParams p;
    p.x = x; p.w = w_packed; p.scale = scale; p.shift = shift; p.y = y;
    p.N = N; p.T = T; p.H = H; p.W = W;
    p.To = (T + 5 - 7) / 2 + 1; p.Ho = (H + 5 - 7) / 2 + 1; p.Wo = (W + 5 - 7) / 2 + 1;
    if (p.To <= 0 || p.Ho <= 0 || p.Wo <= 0) return STEP_E_SHAPE;
    p.Cout = Cout; p.y_cstride = y_cstride; p.y_coff = y_coff;
    p.tiles_h = ceil_div(p.Ho, STEM_TH); p.tiles_w = ceil_div(p.Wo, STEM_TW);
    p.nblk32 = ceil_div(Cout, 32);
    const int ov = conv_impl_override();
    switch (dtype) {
        case STEP_F32: return stem_forward_t<float>(p, stream);
        // STEP_CONV_IMPL=igemm / tap select the two older stems (A/B measurements, tests); default = dense-K stream stem
        case STEP_BF16: return ov == 0 ? stem_forward_t<bf16_t>(p, stream) : (ov == 1 ? stem_tap_forward_t<bf16_t>(p, stream) : stem_stream_forward_t<bf16_t>(p, stream));
        case STEP_F16: return ov == 0 ? stem_forward_t<f16_t>(p, stream) : (ov == 1 ? stem_tap_forward_t<f16_t>(p, stream) : stem_stream_forward_t<f16_t>(p, stream));
    }
    return STEP_E_DTYPE;
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

int step_stem_kernel_name(int dtype, char* buf, int buflen) {
    if (!buf || buflen <= 0) return STEP_E_NULL;
    const int ov = conv_impl_override();
    const char* t = dtype == STEP_F32 ? "float" : (dtype == STEP_BF16 ? "step::bf16_t" : "step::f16_t");
    if (dtype != STEP_F32 && dtype != STEP_BF16 && dtype != STEP_F16) return STEP_E_DTYPE;
    if (dtype == STEP_F32 || ov == 0) snprintf(buf, (size_t)buflen, "void step::stem_igemm_kernel<%s, 2>(step::StemParams)", t);
    else snprintf(buf, (size_t)buflen, "void step::%s<%s>(step::StemParams)", ov == 1 ? "stem_tap_kernel" : "stem_stream_kernel", t);
    return STEP_OK;
}

int step_conv_kernel_name(const step_conv_desc* d, char* buf, int buflen) {
    if (!d || !buf || buflen <= 0) return STEP_E_NULL;
    const step_conv_desc canon = canonical_desc(d);
    d = &canon;
    const ConvPlan pl = conv_plan(d);
    if (!pl.ok) return STEP_E_UNSUPPORTED;
    const char* t = d->dtype == STEP_F32 ? "float" : (d->dtype == STEP_BF16 ? "step::bf16_t" : "step::f16_t");
    if (pl.impl == 3)
        snprintf(buf, (size_t)buflen, "void step::pw_splitk_kernel<%s, %d>(step::ConvParams, float*, int, int, int)", t, pl.mbk);
    else if (pl.impl == 2)
        snprintf(buf, (size_t)buflen, "void step::conv_pw_kernel<%s, %d>(step::ConvParams)", t, pl.NB);
    else if (pl.impl == 1)
        snprintf(buf, (size_t)buflen, "void step::conv_tap_kernel<%s, %d, %d, %d, %d, %d, %d, %d>(step::ConvParams)", t,
                 pl.twl, pl.NB, d->kd, d->kh, d->kw, pl.mb == 4 ? 2 : pl.tps, pl.mb);
    else
        snprintf(buf, (size_t)buflen, "void step::conv_igemm_kernel<%s, %d, %d, %d, %d, %d, %s, %d>(step::ConvParams)", t,
                 pl.flat ? 4 : (pl.wide ? 5 : 4), pl.NB, d->kd, d->kh, d->kw, pl.flat ? "true" : "false", pl.deep ? 128 : 32);
    return STEP_OK;
}

const char* step_version(void) { return "step_amd 0.1.0 gfx950"; }
int step_abi_version(void) { return 6; }

}  // extern "C"
