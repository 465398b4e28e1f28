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
//
// This unit holds the 4-wave conv_igemm_kernel, the weight packer, the launch planner and the step_conv_* entry
// points; the 8-wave kernels live in conv_tap_*.hip / conv_pw.hip, the stem in stem.hip, the weight gradients in
// conv_wgrad.hip (shared declarations: conv_common.h).
#include "conv_common.h"

namespace step {

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


// ---- weight packing: torch [Cout][Cin][taps] fp32 -> [nb32][tap][kc16][lane][8] of T ----------
// DGRAD: pack the weight of the DATA-GRADIENT conv straight from the forward weight w [Cw][Cout][taps] (Cw = the forward
// conv's output channels): effective weight [Cout][Cw][taps] with every tap axis flipped (= the linear tap index
// reversed), input channels >= Cw (the caller's 16-byte padding) zero.
// one element of the packed image (idx = its linear index): the effective weight is  w[:, cin_lo + perm[j]]  of the parameter
// w [.][w_cin][taps]
template <bool DGRAD>
__device__ __forceinline__ float pack_value(const float* __restrict__ w, const int32_t* __restrict__ perm, long long idx, int Cout,
                                            int Cin, int ntaps, int KC16, int Cw, int w_cin, int cin_lo) {
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
    if (DGRAD) {
        if (co < Cout && ci < Cw && tap < ntaps) {
            const int cs = cin_lo + (perm ? perm[co] : co);
            v = w[((size_t)ci * w_cin + cs) * ntaps + (ntaps - 1 - tap)];
        }
    } else if (co < Cout && ci < Cin && tap < ntaps) {
        const int cs = cin_lo + (perm ? perm[ci] : ci);
        v = w[((size_t)co * w_cin + cs) * ntaps + tap];
    }
    return v;
}

template <typename T, bool DGRAD>
__global__ void pack_weight_kernel(const float* __restrict__ w, const int32_t* __restrict__ perm, T* __restrict__ out,
                                   int Cout, int Cin, int ntaps, int KC16, long long total, int Cw) {
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)blockDim.x * gridDim.x)
        out[idx] = elem<T>::from_f32(pack_value<DGRAD>(w, perm, idx, Cout, Cin, ntaps, KC16, Cw, DGRAD ? Cout : Cin, 0));
}

// every weight of a net in one launch: blockIdx.y = the item, blockIdx.x strides over its packed image
template <typename T>
__global__ void pack_weights_kernel(const step_pack_item* __restrict__ items) {
    const step_pack_item it = items[blockIdx.y];
    const int ntaps = it.kd * it.kh * it.kw;
    // the packed conv: forward Cout x Cin; data gradient (channel roles swapped) Cin x cin_pad
    const int pc_out = it.dgrad ? it.Cin : it.Cout, pc_in = it.dgrad ? it.cin_pad : it.Cin;
    const int KC16 = (pc_in + CK - 1) / CK * 2;
    const long long total = (long long)((pc_out + 31) / 32) * taps_padded(ntaps) * KC16 * 512;
    T* out = (T*)it.packed;
    if (it.dgrad) {
        for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)blockDim.x * gridDim.x)
            out[idx] = elem<T>::from_f32(pack_value<true>(it.w, it.perm_c, idx, pc_out, pc_in, ntaps, KC16, it.Cout, it.w_cin, it.cin_lo));
    } else {
        for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)blockDim.x * gridDim.x)
            out[idx] = elem<T>::from_f32(pack_value<false>(it.w, it.perm_c, idx, pc_out, pc_in, ntaps, KC16, 0, it.w_cin, it.cin_lo));
    }
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


static int pick_nb_tap(int nblk32, long long mtiles, int slots = 256) {
    const int forced = opt(STEP_OPT_CONV_NB);             // tests / A-B timing
    if (forced >= 1 && forced <= 3) return forced;
    int best = 1;
    double best_cost = -1;
    for (int nb = 3; nb >= 1; --nb) {   // 2 fragment sets + 2*nb accumulators must fit 256 VGPRs: nb <= 3
        const long long groups = ceil_div(nblk32, 2 * nb);
        const long long wgs = groups * mtiles;
        // `slots` resident workgroups on the chip (256 eight-wave ones, 512 four-wave ones): rounds of that many;
        // per-workgroup time ~ 2*nb MFMA units + fixed part
        // (STEP_OPT_CONV_NB_RULE = 1: the CHIP TIME of the launch -- workgroups x per-workgroup time, no rounding to rounds: with two
        // batches in flight the CUs a one-round launch leaves idle run the other batch, so fewer, deeper workgroups -- more MFMAs per
        // fragment read -- are what counts, not the length of the round)
        const double cost = opt(STEP_OPT_CONV_NB_RULE) == 1 ? (double)wgs * (2.0 * nb + 1.5) - 0.01 * nb
                                                            : (double)ceil_div64(wgs, slots) * (2.0 * nb + 1.5) - 0.01 * nb;
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = nb; }
    }
    return best;
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
static long long best_gen_box(int D, int H, int W, int kd, int npix_limit, int* btd, int* bth, int* btw, int pxmax = 256) {
    long long best = -1; int bhalo = 0;
    for (int kw_ = 1; kw_ <= 8; ++kw_) {
        const int tw = ceil_div(W, kw_);
        if (tw > pxmax) continue;
        if (kw_ > 2 && tw < 16) break;
        for (int kh_ = 1; kh_ <= H; ++kh_) {
            const int th = ceil_div(H, kh_);
            if (th * tw > pxmax) continue;
            if (kh_ > 1 && th == ceil_div(H, kh_ - 1)) continue;
            for (int td = 1; td <= D && td * th * tw <= pxmax; ++td) {
                const int halo = (td + kd - 1) * (th + 2) * (tw + 2);
                if (halo > npix_limit) break;
                const long long tiles = (long long)ceil_div(D, td) * ceil_div(H, th) * ceil_div(W, tw);
                if (best < 0 || tiles < best || (tiles == best && halo < bhalo)) { best = tiles; bhalo = halo; *btd = td; *bth = th; *btw = tw; }
            }
        }
    }
    return best;
}

// Which form of conv_tap_kernel: measured rule (DESIGN.md, 3x3x3 family).
// Interleaved A/B on MI355X over the C2 layers (tools/ab_bench.py, profiles/r02_ab_waves.txt): the four-wave form is slower
// on every 3x3x3 layer that fills the chip (conv3d_2c 290 -> 301 us, 3c_b1b 144 -> 326 us: one tap per barrier and twice
// the weight staging per MFMA cost more than the second resident workgroup hides) and ~7 % faster only on the smallest
// 14x14 branch_2 layers, whose eight-wave launches are a few dozen workgroups.
static bool prefer_four_waves(const ConvPlan& p8, const ConvPlan& p4, const step_conv_desc* d) {
    (void)p4;
    const long long wgs8 = p8.mtiles * ceil_div(ceil_div(d->Cout, 32), 2 * p8.NB);
    return wgs8 < 64;
}

// The streaming pointwise GEMM: four waves (128-pixel tiles, two resident workgroups) measured 6-9 % faster on the 28x28
// layers (3b / 3c fused triples 27.2 -> 24.8 us, 46.6 -> 42.9 us) and 20-25 % faster where the eight-wave grid is below
// half the chip (4b / 4f branch_3: 9.9 -> 7.3 us, 10.8 -> 8.4 us); equal or 1-4 % slower on the 14x14 fused triples.
static bool prefer_four_waves_pw(const step_conv_desc* d, long long wgs8) {
    const long long M = (long long)d->N * d->D * d->H * d->W;
    return wgs8 < 128 || M >= 65536;
}

// general boxes: may every 16-lane LDS read group be one run of 16 columns of one box row (conv_tap_kernel.h, p.gmode)?  Only for
// widths just below a multiple of 16 (the padded lanes cost no more than the linear packing leaves unused) and when the box rows
// fit the tile's 16-lane slots.  STEP_OPT_CONV_GMODE = 0 keeps the linear walk (tests / A-B timing).
static int gen_gmode(int td, int th, int tw, int tile_px) {
    if (opt(STEP_OPT_CONV_GMODE) == 0) return 0;
    const int spr = (tw + 15) / 16;
    if ((tw & 15) < 12) return 0;
    return td * th * spr <= tile_px / 16 ? 1 : 0;
}

static ConvPlan conv_plan(const step_conv_desc* d, bool allow_pws = true, int force_waves = 0, bool no_gen = false) {
    ConvPlan pl;
    pl.ok = true; pl.wide = false; pl.tiles_h = pl.tiles_w = 0; pl.impl = 0; pl.tps = 1; pl.mb = 2; pl.wv = 8; pl.ph = 0; pl.deep = false; pl.twl = 4; pl.tiles_d = d->D; pl.gtd = pl.gth = pl.gtw = 1; pl.gmode = 0; pl.ksplit = 0; pl.kchunk16 = 0; pl.mbk = 0; pl.mpad = 0; pl.cpad = 0;
    const int nblk32 = ceil_div(d->Cout, 32);
    const bool k1 = d->kd == 1 && d->kh == 1 && d->kw == 1;
    const bool k333 = d->kd == 3 && d->kh == 3 && d->kw == 3;
    const bool k133 = d->kd == 1 && d->kh == 3 && d->kw == 3;
    pl.flat = k1;
    if (k1) {
        pl.mtiles = ceil_div64((long long)d->N * d->D * d->H * d->W, 128);
        pl.NB = pick_nb(nblk32, pl.mtiles);
        // deep-K pointwise convs on enough pixels: the streaming 8-wave GEMM (STEP_OPT_CONV_IMPL = 0 forces the other)
        const long long mt256 = ceil_div64((long long)d->N * d->D * d->H * d->W, 256);
        const int ov1 = conv_impl_override();
        {
            // the weight-stationary stream (conv_pws_kernel): 16-bit, 16-byte channel vectors, the weights of a channel group
            // (NB blocks x K) within 104 KiB of LDS.  STEP_OPT_CONV_PWS: 1 = wherever the contract allows, 0 = never.
            const int pws_env = opt(STEP_OPT_CONV_PWS);
            const int KC16 = ceil_div(d->Cin, CK) * 2;
            const long long M = (long long)d->N * d->D * d->H * d->W;
            const bool can = d->dtype != STEP_F32 && (d->Cin % 8) == 0 && (d->x_cstride % 8) == 0 && (d->x_coff % 8) == 0 && KC16 <= 16 &&
                             (d->y_cstride % 8) == 0 && (d->y_coff % 8) == 0 && (d->Cout % 8) == 0 && (d->res_cstride % 8) == 0 && (d->res_coff % 8) == 0 &&
                             (d->split == 0 || ((d->split % 8) == 0 && (d->y2_cstride % 8) == 0 && (d->y2_coff % 8) == 0)) && M >= 1024;
            // measured (tools/ab_bench.py, bf16): it wins with K = 192 .. 256 and many channel blocks on a large map -- the 3c fused
            // triple 43.1 -> 37.4 us at 28x28 (batch 8), 65.2 -> 55.3 us at 50x50 (batch 4), and with the 16-byte register epilogue
            // (round 3) 34.2 / 48.7 us; the 3b triple (K = 192) 26.1 -> 24.3 us since that epilogue; equal or slower on conv3d_2b
            // and the narrow branch_3 layers: the default covers that class only
            // with a residual (round 6, the heads' Bottleneck conv3 256 -> 1024 at the reference's 34 tubes per clip: 59 976 rows) 124 -> 107 us
            // (8-byte residual loads) -> see profiles/r06_ab_pws_heads.txt for the 16-byte form; 19 992 rows: 38 -> 42 us, and 256 -> 256 a tie
            // at best: wide outputs on many rows only
            const bool wins = d->res_cstride == 0 ? (KC16 >= 12 && nblk32 >= 6 && M >= 65536) : (KC16 >= 12 && nblk32 >= 16 && M >= 49152);
            if (allow_pws && can && pws_env != 0 && (pws_env == 1 || wins) && ov1 == -1) {
                int nbmax = 152 / (ceil_div(KC16, 4) * 4);               // (LDS holds K padded to whole 64-channel steps)
                if (nbmax > 16) nbmax = 16;
                if (nbmax > nblk32) nbmax = nblk32;
                const int groups = ceil_div(nblk32, nbmax);
                pl.impl = 4;
                pl.NB = ceil_div(nblk32, groups);                         // channel blocks per workgroup (balanced groups)
                pl.wv = 8;
                long long gx = 256 / groups;                               // one workgroup per CU
                if (gx < 1) gx = 1;
                const long long need = ceil_div64(ceil_div64(M, 32), 8);
                if (gx > need) gx = need;
                pl.mtiles = gx;
                return pl;
            }
        }
        if (ov1 == 5 || (ov1 != 0 && d->Cin >= 128 && d->Cout >= 64 && mt256 * ceil_div(nblk32, 2) >= 32)) {
            pl.impl = 2;
            const int waves_env = opt(STEP_OPT_CONV_WAVES);      // tests / A-B timing: 4 | 8
            const long long wgs8 = mt256 * ceil_div(nblk32, 2 * pick_nb_tap(nblk32, mt256));
            const bool four = waves_env == 4 || (waves_env != 8 && prefer_four_waves_pw(d, wgs8));
            pl.wv = four ? 4 : 8;
            pl.mtiles = four ? ceil_div64((long long)d->N * d->D * d->H * d->W, 128) : mt256;
            pl.NB = pick_nb_tap(nblk32, pl.mtiles, four ? 512 : 256);
            return pl;
        }
        // few rows x very deep K (the heads' Linear layers): split K over the chip (needs the workspace of
        // step_conv_forward_ws; STEP_OPT_CONV_SPLITK = 0 disables)
        const bool splitk_ok = opt(STEP_OPT_CONV_SPLITK) != 0;
        const long long M = (long long)d->N * d->D * d->H * d->W;
        // rows: up to 4096 (round 6; 1024 until then).  The reference's default of 34 tubes per clip makes the last step's Linear layers 4 x 34 x 9 =
        // 1224 rows: just past the old bound they fell onto the tiled kernel -- 20 workgroups walking K = 12544 one slab after the other, 233 us a
        // launch, three launches = 11 % of the C3 step at 34 tubes (profiles/r06_c3_34_kernel_stats.txt)
        // (the wider bound only where no streaming form exists, Cout < 64: a backbone layer such as 528 -> 128 on ONE clip's 1568 rows would otherwise
        // change its K summation order with the batch size -- case_c2_full_size_properties holds a clip's features bit-identical across batches)
        if (splitk_ok && ov1 != 0 && d->Cin >= 512 && (d->Cin % 8) == 0 && (M <= 1024 || (M <= 4096 && d->Cout < 64)) && pl.mtiles * nblk32 < 64) {     // (>= 512: the 1024-channel context half of global_cls on ~130 rows ran 37 us as two serial workgroups)
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
    // STEP_OPT_CONV_GEN = <percent> moves the threshold, 0 disables the general boxes.
    int gtd = 1, gth = 1, gtw = 1;
    const int gen_pct = no_gen ? 0 : opt(STEP_OPT_CONV_GEN);    // (no_gen: the caller wants a power-of-two tile -- the pooled epilogue of step_conv_forward_pre_pool)
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
    // ---- the four-wave form (128-pixel tiles, two resident workgroups per CU; conv_tap_kernel.h): its own tile search
    const int waves_env = force_waves ? force_waves : opt(STEP_OPT_CONV_WAVES);      // tests / A-B timing: 4 | 8
    ConvPlan p4 = pl;
    bool have4 = false;
    {
        const long long a16 = (long long)d->D * ceil_div(d->H, 8) * ceil_div(d->W, 16);
        const long long a32 = (long long)d->D * ceil_div(d->H, 4) * ceil_div(d->W, 32);
        const long long a8 = (long long)ceil_div(d->D, 2) * ceil_div(d->H, 8) * ceil_div(d->W, 8);
        int tw4 = 4;
        long long tb4 = a16;
        if (a8 < tb4) { tb4 = a8; tw4 = 3; }
        // (the 4 x 32 shape only when it beats both: its 612-pixel halo does not leave room for a second workgroup)
        if (a32 * 100 < tb4 * 90) { tb4 = a32; tw4 = 5; }
        int g4d = 1, g4h = 1, g4w = 1;
        if (gen_pct > 0) {
            int limit = CONV_GEN_NPIX4_WIDE;
            for (int pass = 0; pass < 2; ++pass) {
                int td_ = 1, th_ = 1, tw_ = 1;
                const long long tg = best_gen_box(d->D, d->H, d->W, d->kd, limit, &td_, &th_, &tw_, 128);
                if (!(tg > 0 && tg * 100 <= tb4 * gen_pct)) break;
                const int nb = pick_nb_tap(nblk32, (long long)d->N * tg, 512);
                if (nb >= 3 && limit != CONV_GEN_NPIX4) { limit = CONV_GEN_NPIX4; continue; }    // NB = 3 reserves a smaller halo
                tb4 = tg; tw4 = 0; g4d = td_; g4h = th_; g4w = tw_;
                break;
            }
        }
        p4.wv = 4; p4.twl = tw4; p4.wide = tw4 == 5; p4.gtd = g4d; p4.gth = g4h; p4.gtw = g4w;
        p4.gmode = tw4 == 0 ? gen_gmode(g4d, g4h, g4w, 128) : 0;
        if (tw4 == 0) {
            p4.tiles_d = ceil_div(d->D, g4d); p4.tiles_h = ceil_div(d->H, g4h); p4.tiles_w = ceil_div(d->W, g4w);
        } else {
            p4.tiles_d = tw4 == 3 ? ceil_div(d->D, 2) : d->D;
            p4.tiles_h = tw4 == 5 ? ceil_div(d->H, 4) : ceil_div(d->H, 8);
            p4.tiles_w = tw4 == 4 ? ceil_div(d->W, 16) : (tw4 == 5 ? ceil_div(d->W, 32) : ceil_div(d->W, 8));
        }
        p4.mtiles = (long long)d->N * tb4;
        p4.NB = pick_nb_tap(nblk32, p4.mtiles, 512);
        if (tw4 == 0 && p4.NB >= 3 && (g4d + d->kd - 1) * (g4h + 2) * (g4w + 2) > CONV_GEN_NPIX4) p4.NB = 2;   // (forced NB with a wide box)
        p4.impl = 1; p4.tps = p4.NB == 1 ? 2 : 1; p4.mb = 2;
        have4 = true;
    }
    // few-tile, small-Cin problems stay on the 4-wave 128-pixel kernel (more workgroups); everything else -- the
    // Cin = 16/32 branches of the Inception blocks included (measured: 9 x 3x3x3 layers 0.253 ms against 0.317 ms) --
    // runs the pipelined kernel
    // (conv_tap_kernel keeps 32-bit element offsets of its halo vectors)
    const bool fits32 = ((unsigned long long)d->N * d->D * d->H * d->W + 1) * (unsigned long long)d->x_cstride < 0xffffffffULL;
    const bool use_tap = fits32 && (ov >= 1 || (ov != 0 && (d->Cin >= 64 || mt256 >= 32)));
    if (use_tap) {
        pl.impl = 1;
        pl.tps = (ov == 1) ? 1 : 2;     // two taps per barrier measured 6-15 % faster than one (STEP_OPT_CONV_IMPL = 1 forces one)
        pl.mb = 2;                      // (the 4-wave MB = 4 form of the kernel template spills at NB >= 2 and is not instantiated)
        pl.twl = twl;
        pl.wide = twl == 5;
        pl.gtd = gtd; pl.gth = gth; pl.gtw = gtw;
        pl.gmode = twl == 0 ? gen_gmode(gtd, gth, gtw, 256) : 0;
        if (twl == 0) {
            pl.tiles_d = ceil_div(d->D, gtd); pl.tiles_h = ceil_div(d->H, gth); pl.tiles_w = ceil_div(d->W, gtw);
        } else {
            pl.tiles_d = twl == 3 ? ceil_div(d->D, 4) : d->D;
            pl.tiles_h = twl == 4 ? ceil_div(d->H, 16) : ceil_div(d->H, 8);
            pl.tiles_w = twl == 4 ? ceil_div(d->W, 16) : (twl == 5 ? ceil_div(d->W, 32) : ceil_div(d->W, 8));
        }
        pl.mtiles = mt256;
        pl.NB = pick_nb_tap(nblk32, pl.mtiles);
        // the two-phase form (anti-phase wave groups, conv_tap_kernel.h): 16-bit storage, two taps per step
        // (measured on MI355X, C2 layers, interleaved A/B, profiles/r02_ab_phased.txt: every 3x3x3 layer faster, 829 -> 704 us
        // per step in total; conv3d_2c 299 -> 259 us, the 14x14 branch_1 layers -19...-23 %)
        if (d->dtype != STEP_F32 && pl.tps == 2 && k333) {
            pl.ph = opt(STEP_OPT_CONV_PHASED) ? 1 : 0;            // tests / A-B timing: 0 = the classic form
        }
        // ... and the 1x3x3 windows of the heads on general boxes at NB <= 2 (round 6: launch_tap_ph_133, conv_tap_kernel.h).  Measured on the C3
        // pipeline, variants alternating on one box (profiles/r06_ab_k133_two_phase.txt): 4 clips x 11 tubes 1 035-1 044 -> 1 053-1 055 clips/s
        // (+1.4 %), x 34 tubes 657-663 -> 662-664, the C4 step unchanged.  STEP_OPT_CONV_PHASED = 1 keeps them on the classic form (A/B).
        if (d->dtype != STEP_F32 && pl.tps == 2 && k133 && pl.twl == 0 && pl.NB <= 2 && opt(STEP_OPT_CONV_PHASED) == 2) pl.ph = 1;
        if (have4 && ov != 1 && (waves_env == 4 || (waves_env != 8 && prefer_four_waves(pl, p4, d)))) return p4;
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


// reciprocals of a general box's halo extents for the kernels' index tables (conv_common.h: mag_hhw / mag_hw)
static inline void conv_set_box_magic(ConvParams& p, int kh, int kw) {
    const unsigned long long hw = (unsigned long long)(p.gtw + kw - 1), hhw = hw * (unsigned long long)(p.gth + kh - 1);
    p.mag_hw = (unsigned)((0x100000000ULL + hw - 1) / hw);
    p.mag_hhw = (unsigned)((0x100000000ULL + hhw - 1) / hhw);
}

template <typename T>
static int conv_forward_t(const step_conv_desc* d, ConvParams p, void* ws, size_t ws_bytes, step_stream_t stream) {
    constexpr int VEC = elem<T>::VEC;
    if (d->Cin % VEC || d->x_cstride % VEC || d->x_coff % VEC) return STEP_E_ALIGN;
    if (((uintptr_t)p.x % 16) || ((uintptr_t)p.w % 16)) return STEP_E_ALIGN;
    ConvPlan pl = conv_plan(d, p.x2 == nullptr, 0, p.pool_row != nullptr && p.pool_p2);
    if (!pl.ok) return STEP_E_UNSUPPORTED;
    if (p.x2 && pl.impl != 2) return STEP_E_UNSUPPORTED;        // (two sources: the streaming GEMM only; the caller launches the halves one after the other)
    if (p.x2 && pl.wv == 8 && pl.NB == 3) pl.NB = 2;            // (conv_pw2_kernel has no eight-wave NB = 3 form: registers)
    if (pl.impl == 3) {
        const size_t need = (size_t)pl.ksplit * pl.mpad * pl.cpad * sizeof(float);
        if (ws && ws_bytes >= need && ((uintptr_t)ws % 16) == 0) return conv_splitk_launch<T>(pl, p, (float*)ws, stream);
        // no workspace (plain step_conv_forward): the tiled kernel
        pl.impl = 0; pl.mtiles = ceil_div64(p.Mtot, 128); pl.NB = pick_nb(p.nblk32, pl.mtiles); pl.deep = d->Cin >= 256;
    }
    p.tiles_h = pl.tiles_h; p.tiles_w = pl.tiles_w; p.tiles_d = pl.tiles_d;
    p.gtd = pl.gtd; p.gth = pl.gth; p.gtw = pl.gtw; p.gmode = pl.gmode;
    conv_set_box_magic(p, d->kh, d->kw);
    auto grid1d = [&](int groups) {                   // logical (mtiles x groups) grid as a 1-D launch padded to 8
        p.gx = (int)pl.mtiles; p.gy = groups;
        const long long tot = pl.mtiles * groups;
        p.gbase = 0; p.gcount = (int)((tot + 7) / 8 * 8);
        // the persistent tile loop (conv_tap_kernel.h, PERSIST): one channel group, more than one round of the chip's one-per-CU slots --
        // today the fused conv3d_2b -> 2c -> pool call (STEP_OPT_CONV_PERSIST = 0: one workgroup per tile; bit-identical)
        p.gpersist = 0;
        const long long slots_ = opt(STEP_OPT_CONV_SLOTS) > 0 ? opt(STEP_OPT_CONV_SLOTS) : 256;
        // (measured with two batches in flight too: C2 1.0674 / 1.0708 ms without against 1.0696 / 1.0680 ms with -- inside the noise -- and
        // C5, whose conv3d_2c is 21 rounds, 3.4772 -> 3.4168 ms: -1.7 %; gpurun_out/ab_c2.txt, ab_c15_*.txt)
        if (opt(STEP_OPT_CONV_PERSIST) != 0 && groups == 1 && pl.impl == 1 && pl.ph == 1 && pl.NB == 3 && p.pre_w && p.pool_row && tot > slots_)
            p.gpersist = (int)(slots_ / 8 * 8 > 0 ? slots_ / 8 * 8 : 8);
        return dim3((unsigned)p.gcount);
    };
    if (pl.impl == 4) {
        if (p.vec_epi && (!p.res || ((p.r_cstride % 8) == 0 && (p.r_coff % 8) == 0 && ((uintptr_t)p.res % 16) == 0)))
            return conv_pws_launch<T>(pl.NB, p, dim3((unsigned)pl.mtiles, (unsigned)ceil_div(p.nblk32, pl.NB)), stream);
        pl = conv_plan(d, false);                               // (an output pointer off the 16-byte grid: the general kernels)
    }
    if (pl.impl == 2) {
        dim3 grid = grid1d(ceil_div(p.nblk32, 2 * pl.NB));
        return conv_pw_launch<T>(pl.NB, pl.wv, p, grid, stream);
    }
    if (pl.impl == 1) {
        // The last, partial round of one-workgroup-per-CU tiles: conv3d_2c at C2 is 1568 tiles = 6.125 rounds of 256, i.e. a
        // seventh round that keeps 32 CUs busy and 224 idle for a full tile time (12 % of the launch).  When the layer is ONE
        // channel group deep (NB > 1) those tail tiles are launched separately with NB = 1 -- NB times as many, shorter
        // workgroups spread over the idle CUs.  Same pixels, same K order per output: bit-identical.  (STEP_OPT_CONV_TAIL = 0: one launch.)
        const int groups = ceil_div(p.nblk32, 2 * pl.NB);
        const long long slots = opt(STEP_OPT_CONV_SLOTS) > 0 ? opt(STEP_OPT_CONV_SLOTS) : 256;     // (tests: the split at interpreter sizes)
        const long long tail = pl.mtiles % slots;
        const bool tail_ok = opt(STEP_OPT_CONV_TAIL) != 0;
        const int tgroups = ceil_div(p.nblk32, 2);
        const bool box_ok = pl.twl != 0 || (pl.gtd + d->kd - 1) * (pl.gth + d->kh - 1) * (pl.gtw + d->kw - 1) <= conv_gen_npix(8, 1);   // (the NB = 1 kernel reserves a smaller general-box halo)
        if (tail_ok && box_ok && pl.wv == 8 && groups == 1 && pl.NB > 1 && pl.mtiles > slots && tail > 0 && tail * 4 <= slots && tail * tgroups <= slots) {
            const long long all = pl.mtiles;
            pl.mtiles = all - tail;
            dim3 grid = grid1d(1);
            int rc = conv_tap_launch<T>(pl, p, d->kd, grid, stream);
            if (rc != STEP_OK) return rc;
            ConvPlan pt = pl;
            pt.NB = 1; pt.mtiles = tail;
            p.gpersist = 0;
            p.tile0 = (int)(all - tail);
            p.gx = (int)tail; p.gy = tgroups;
            p.gcount = (int)((tail * tgroups + 7) / 8 * 8);
            const dim3 gt((unsigned)p.gcount);
            return conv_tap_launch<T>(pt, p, d->kd, gt, stream);
        }
        dim3 grid = grid1d(groups);
        return conv_tap_launch<T>(pl, p, d->kd, grid, stream);
    }
    dim3 grid = grid1d(ceil_div(p.nblk32, pl.NB));
    if (pl.flat)
        return pl.deep ? launch_nb<T, 4, 1, 1, 1, true, 128>(p, pl.NB, grid, stream) : launch_nb<T, 4, 1, 1, 1, true, 32>(p, pl.NB, grid, stream);
    if (d->kd == 3)
        return pl.wide ? launch_nb<T, 5, 3, 3, 3, false, 32>(p, pl.NB, grid, stream) : launch_nb<T, 4, 3, 3, 3, false, 32>(p, pl.NB, grid, stream);
    return pl.wide ? launch_nb<T, 5, 1, 3, 3, false, 32>(p, pl.NB, grid, stream) : launch_nb<T, 4, 1, 3, 3, false, 32>(p, pl.NB, grid, stream);
}


}  // namespace step

using namespace step;

#ifdef STEP_PROBE
unsigned long long* step::g_probe_buf = nullptr;
#endif

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
        case STEP_F32: STEP_LAUNCH((pack_weight_kernel<float, false>), grid, dim3(256), stream, w, perm, (float*)packed, Cout, Cin, ntaps, KC16, total, 0); break;
        case STEP_BF16: STEP_LAUNCH((pack_weight_kernel<bf16_t, false>), grid, dim3(256), stream, w, perm, (bf16_t*)packed, Cout, Cin, ntaps, KC16, total, 0); break;
        case STEP_F16: STEP_LAUNCH((pack_weight_kernel<f16_t, false>), grid, dim3(256), stream, w, perm, (f16_t*)packed, Cout, Cin, ntaps, KC16, total, 0); break;
        default: return STEP_E_DTYPE;
    }
    return STEP_LAUNCH_CHECK();
}

int step_conv_pack_weight_dgrad(const float* w, int Cout, int Cin, int kd, int kh, int kw, int dtype, int cin_pad, void* packed,
                                step_stream_t stream) {
    // w [Cout][Cin][taps] = the FORWARD weight; the packed image is the one step_conv_forward wants for the conv
    // gy [.., cin_pad >= Cout] -> gx [.., Cin]
    if (Cout <= 0 || Cin <= 0 || kd <= 0 || kh <= 0 || kw <= 0 || cin_pad < Cout) return STEP_E_SHAPE;
    if (!(kd & 1) || !(kh & 1) || !(kw & 1)) return STEP_E_UNSUPPORTED;       // SAME padding is symmetric for odd kernels only
    if (!w || !packed) return STEP_E_NULL;
    const long long total = (long long)step_conv_packed_elems(Cin, cin_pad, kd, kh, kw);
    const int KC16 = ceil_div(cin_pad, CK) * 2, ntaps = kd * kh * kw;
    const dim3 grid(flat_grid(total, 256));
    switch (dtype) {
        case STEP_F32: STEP_LAUNCH((pack_weight_kernel<float, true>), grid, dim3(256), stream, w, nullptr, (float*)packed, Cin, cin_pad, ntaps, KC16, total, Cout); break;
        case STEP_BF16: STEP_LAUNCH((pack_weight_kernel<bf16_t, true>), grid, dim3(256), stream, w, nullptr, (bf16_t*)packed, Cin, cin_pad, ntaps, KC16, total, Cout); break;
        case STEP_F16: STEP_LAUNCH((pack_weight_kernel<f16_t, true>), grid, dim3(256), stream, w, nullptr, (f16_t*)packed, Cin, cin_pad, ntaps, KC16, total, Cout); break;
        default: return STEP_E_DTYPE;
    }
    return STEP_LAUNCH_CHECK();
}


int step_conv_pack_weights(const step_pack_item* items, int n, int dtype, step_stream_t stream) {
    // the descriptors live on the device: shapes are validated by the caller's binding (step_amd/ops.py) -- here only what
    // the launch itself needs
    if (n < 0) return STEP_E_SHAPE;
    if (n == 0) return STEP_OK;
    if (!items) return STEP_E_NULL;
    const dim3 grid(48, (unsigned)n);
    switch (dtype) {
        case STEP_F32: STEP_LAUNCH((pack_weights_kernel<float>), grid, dim3(256), stream, items); break;
        case STEP_BF16: STEP_LAUNCH((pack_weights_kernel<bf16_t>), grid, dim3(256), stream, items); break;
        case STEP_F16: STEP_LAUNCH((pack_weights_kernel<f16_t>), grid, dim3(256), stream, items); break;
        default: return STEP_E_DTYPE;
    }
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

// argument checks + the kernel parameter block of one conv; `canon` receives the canonical descriptor.  STEP_OK with p.N == 0: nothing to launch.
static int conv_fill_params(const step_conv_desc* d, const void* x, const void* w_packed, const float* scale, const float* shift,
                            const void* res, void* y, void* y2, step_conv_desc& canon, ConvParams& p) {
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
    p.N = d->N;
    if (d->N == 0) return STEP_OK;
    if (!x || !w_packed || !y) return STEP_E_NULL;
    canon = canonical_desc(d);
    d = &canon;
    p.x = x; p.w = w_packed; p.scale = scale; p.shift = shift; p.res = res; p.y = y; p.y2 = y2;
    p.split = split; p.y2_cstride = d->y2_cstride; p.y2_coff = d->y2_coff;
    p.N = d->N; p.D = d->D; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Cout = d->Cout;
    p.x_cstride = d->x_cstride; p.x_coff = d->x_coff; p.y_cstride = d->y_cstride; p.y_coff = d->y_coff;
    p.r_cstride = d->res_cstride; p.r_coff = d->res_coff;
    p.relu = d->relu;
    p.tiles_h = p.tiles_w = 0; p.tiles_d = d->D; p.gtd = p.gth = p.gtw = 1; p.gmode = 0; p.gx = p.gy = 0; p.tile0 = 0; p.gbase = 0; p.gcount = 0; p.gpersist = 0;
    p.x2 = nullptr; p.x2_cstride = p.x2_coff = p.s_split = 0;
    p.nchunks = ceil_div(d->Cin, CK);
    p.nchunks32 = p.nchunks;
    p.vec_epi = (d->y_cstride % 8 == 0) && (d->y_coff % 8 == 0) && (d->Cout % 8 == 0) && (((uintptr_t)y) % 16 == 0) &&
                (!res || ((d->res_cstride % 8 == 0) && (d->res_coff % 8 == 0) && (((uintptr_t)res) % 16 == 0))) &&
                (!split || ((split % 8 == 0) && (d->y2_cstride % 8 == 0) && (d->y2_coff % 8 == 0) && (((uintptr_t)y2) % 16 == 0)));
    p.nblk32 = ceil_div(d->Cout, 32);
    p.pre_w = nullptr; p.pre_scale = nullptr; p.pre_shift = nullptr;
    p.pool_row = nullptr; p.pool_col = nullptr; p.Hp = p.Wp = 0; p.pool_p2 = 0;
    p.mag_hhw = p.mag_hw = 0;
    p.Mtot = (long long)d->N * d->D * d->H * d->W;
#ifdef STEP_PROBE
    p.probe = step::g_probe_buf;
#endif
    return STEP_OK;
}

int step_conv_forward_ws(const step_conv_desc* d, const void* x, const void* w_packed, const float* scale,
                         const float* shift, const void* res, void* y, void* y2, void* ws, size_t ws_bytes, step_stream_t stream) {
    step_conv_desc canon;
    ConvParams p;
    const int rc = conv_fill_params(d, x, w_packed, scale, shift, res, y, y2, canon, p);
    if (rc != STEP_OK || p.N == 0) return rc;
    switch (canon.dtype) {
        case STEP_F32: return conv_forward_t<float>(&canon, p, ws, ws_bytes, stream);
        case STEP_BF16: return conv_forward_t<bf16_t>(&canon, p, ws, ws_bytes, stream);
        case STEP_F16: return conv_forward_t<f16_t>(&canon, p, ws, ws_bytes, stream);
    }
    return STEP_E_DTYPE;
}

// accumulator depth of the pointwise workgroups inside step_pool_conv_forward's grid: 128-channel workgroups (NB = 2) where they fill the
// chip's two-per-CU slots three times over (round 6; option conv_nb = 1 | 2 forces).  Measured on the C3 pipeline, variants alternating on one
// box (profiles/r06_ab_pws_heads.txt, call c42): with the bound at 1024 workgroups the 25x25 backbone maps of 4 clips (1056-1408) went to NB = 2
// as well and the 11-tube line LOST 1.5 % (the pool's workgroups share the arena and drop from three to two per CU with it); the heads' Mixed_5b /
// 5c at 4 x 34 tubes x 9 frames (1876 / 2345 workgroups, GEMM-dominated) gain: 34 tubes +0.6 % even with the backbone's loss inside.
static int pool_conv_nbc(const ConvPlan& pl, long long M, int nblk32) {
    const int nb_env = opt(STEP_OPT_CONV_NB);
    if (nb_env == 1 || nb_env == 2) return nb_env;
    return (pl.NB == 2 && ceil_div64(M, 128) * ceil_div(nblk32, 4) >= 1536) ? 2 : 1;
}

int step_pool_conv_plan_nb(const step_conv_desc* d) {
    if (!d || d->dtype == STEP_F32 || !(d->kd == 1 && d->kh == 1 && d->kw == 1) || d->N <= 0) return 0;
    const ConvPlan pl = conv_plan(d, false);
    if (!pl.ok || pl.impl != 2 || pl.NB > 2) return 0;
    return pool_conv_nbc(pl, (long long)d->N * d->D * d->H * d->W, ceil_div(d->Cout, 32));
}

int step_conv_forward_cat(const step_conv_desc* d, const void* x, int cin_a, const void* xb, int xb_cstride, int xb_coff, const void* w_packed,
                          const float* scale, const float* shift, const void* res, void* y, void* y2, step_stream_t stream) {
    if (!d) return STEP_E_NULL;
    if (cin_a <= 0 || cin_a >= d->Cin) return STEP_E_SHAPE;
    const int cin_b = d->Cin - cin_a;
    if (xb_coff < 0 || xb_coff + cin_b > xb_cstride) return STEP_E_SHAPE;
    step_conv_desc da = *d;
    da.Cin = cin_a;                                             // (the first source's slice is what d's x_cstride / x_coff describe)
    step_conv_desc canon;
    ConvParams p;
    const int rc = conv_fill_params(&da, x, w_packed, scale, shift, res, y, y2, canon, p);
    if (rc != STEP_OK || p.N == 0) return rc;
    if (!xb) return STEP_E_NULL;
    // the form that exists: a 16-bit pointwise conv the planner streams, the first source in whole 32-channel K steps
    if (canon.dtype == STEP_F32 || !(canon.kd == 1 && canon.kh == 1 && canon.kw == 1) || (cin_a % 32) || (cin_b % 8) || (xb_cstride % 8) || (xb_coff % 8))
        return STEP_E_UNSUPPORTED;
    if ((uintptr_t)xb % 16) return STEP_E_ALIGN;
    canon.Cin = d->Cin;
    p.Cin = d->Cin; p.nchunks = ceil_div(d->Cin, CK); p.nchunks32 = p.nchunks;
    p.x2 = xb; p.x2_cstride = xb_cstride; p.x2_coff = xb_coff; p.s_split = cin_a / 32;
    switch (canon.dtype) {
        case STEP_BF16: return conv_forward_t<bf16_t>(&canon, p, nullptr, 0, stream);
        case STEP_F16: return conv_forward_t<f16_t>(&canon, p, nullptr, 0, stream);
    }
    return STEP_E_DTYPE;
}

int step_pool_conv_forward(int dtype, const void* x, int N, int D, int H, int W, int C, int x_cstride, int x_coff, void* pool_y, int py_cstride,
                           int py_coff, const step_conv_desc* d, const void* cx, const void* w_packed, const float* scale, const float* shift,
                           void* y, void* y2, step_stream_t stream) {
    if (N < 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0) return STEP_E_SHAPE;
    if (x_coff < 0 || x_coff + C > x_cstride || py_coff < 0 || py_coff + C > py_cstride) return STEP_E_SHAPE;
    step_conv_desc canon;
    ConvParams p;
    const int rc = conv_fill_params(d, cx, w_packed, scale, shift, nullptr, y, y2, canon, p);
    if (rc != STEP_OK) return rc;
    if (N == 0 && p.N == 0) return STEP_OK;
    if (!x || !pool_y) return STEP_E_NULL;
    // both halves on their combined form only: a 16-bit pointwise conv the planner streams (conv_pw_kernel), run here at NB = 1 with
    // four waves -- other instantiations of the same kernel, the same K order per output: bit-identical to the separate launch
    if (N == 0 || p.N == 0 || canon.dtype != dtype || dtype == STEP_F32 || !(canon.kd == 1 && canon.kh == 1 && canon.kw == 1)) return STEP_E_UNSUPPORTED;
    constexpr int VEC = 8;
    if (canon.Cin % VEC || canon.x_cstride % VEC || canon.x_coff % VEC || ((uintptr_t)p.x % 16) || ((uintptr_t)p.w % 16) || ((uintptr_t)x % 16) ||
        ((uintptr_t)pool_y % 16))
        return STEP_E_UNSUPPORTED;
    // (layers the planner gives the deepest accumulators -- the 28x28 triples, NB = 3 -- keep their own launch: at NB = 1 they would
    // re-read the input three times; NB = 2 layers, mixed_4f's triple, measured +0.3 % riding at NB = 1)
    const ConvPlan pl = conv_plan(&canon, false);
    if (!pl.ok || pl.impl != 2 || pl.NB > 2) return STEP_E_UNSUPPORTED;
    p.tiles_h = pl.tiles_h; p.tiles_w = pl.tiles_w; p.tiles_d = pl.tiles_d;
    p.gtd = pl.gtd; p.gth = pl.gth; p.gtw = pl.gtw; p.gmode = pl.gmode;
    const long long mtiles = ceil_div64(p.Mtot, 128);
    const int nbc = pool_conv_nbc(pl, p.Mtot, p.nblk32);
    const int groups = ceil_div(p.nblk32, 2 * nbc);
    p.gx = (int)mtiles; p.gy = groups;
    const long long tot = (mtiles * groups + 7) / 8 * 8;
    return pool333_pw_launch(dtype, x, N, D, H, W, C, x_cstride, x_coff, pool_y, py_cstride, py_coff, p, tot, nbc, stream);
}

int step_conv_forward_pre(const step_conv_desc* d, const void* x, const void* w_packed, const float* scale, const float* shift,
                          const void* pre_w_packed, const float* pre_scale, const float* pre_shift, int pre_cin, void* y, step_stream_t stream) {
    step_conv_desc canon;
    ConvParams p;
    const int rc = conv_fill_params(d, x, w_packed, scale, shift, nullptr, y, nullptr, canon, p);
    if (rc != STEP_OK) return rc;
    if (!pre_w_packed) return STEP_E_NULL;
    // the fused form exists for what the backbone needs: 64 -> 64 pointwise in front of a 16-bit 3x3x3 conv on the two-phase kernel
    if (pre_cin != 64 || canon.Cin != 64 || canon.dtype == STEP_F32 || canon.split || !(canon.kd == 3 && canon.kh == 3 && canon.kw == 3))
        return STEP_E_UNSUPPORTED;
    if (((uintptr_t)pre_w_packed % 16) || (pre_scale && ((uintptr_t)pre_scale % 16)) || (pre_shift && ((uintptr_t)pre_shift % 16))) return STEP_E_ALIGN;
    if (p.N == 0) return STEP_OK;
    if ((unsigned long long)p.Mtot * (unsigned long long)canon.x_cstride >= 0xffffffffULL) return STEP_E_UNSUPPORTED;      // 32-bit element offsets
    const ConvPlan pl = conv_plan(&canon);
    if (!pl.ok || pl.impl != 1 || pl.ph != 1 || pl.wv != 8 || pl.tps != 2 || (pl.twl != 0 && pl.twl != 3)) return STEP_E_UNSUPPORTED;
    p.pre_w = pre_w_packed; p.pre_scale = pre_scale; p.pre_shift = pre_shift;
    return canon.dtype == STEP_BF16 ? conv_forward_t<bf16_t>(&canon, p, nullptr, 0, stream) : conv_forward_t<f16_t>(&canon, p, nullptr, 0, stream);
}

// ---- conv3d_2b -> conv3d_2c -> maxPool3d_3a as one call (step_conv_forward_pre_pool) ---------------------------------------------------
// Completes the pooled pixels on tile seams of conv_tap_pre_pool_kernel (see the POOL epilogue of conv_tap_body): one thread per (plane,
// seam pixel, 8-channel vector).  PT = pooled pixels per tile side (4: 8 x 8 tiles).  Row seams: pooled row PT*t - 1 lacks conv row
// 2*PT*t = the first row of tile row t (pool_row); the pixels that are ALSO on a column seam take the two pool_col rows they lack here
// too, so that exactly one thread updates any pooled pixel.  Column seams: pooled column PT*s - 1 lacks conv column 2*PT*s (pool_col),
// rows 2ph .. 2ph+2 -- all inside one tile row unless ph is a seam row (handled by the row pass).  Same structure as
// stem_pool_fix_kernel (stem.hip, PT = 8).
}  // extern "C"
namespace step {
struct PoolFixParams { void* y; const void* rowbuf; const void* colbuf; long long planes; int H, W, Hp, Wp, tiles_h, tiles_w, C, y_cstride, y_coff, PT; };
typedef short s16x8_fix __attribute__((ext_vector_type(8)));
__device__ __forceinline__ u32x4 pk_max_nonneg16_fix(const u32x4& a, const u32x4& b) {
    return __builtin_bit_cast(u32x4, __builtin_elementwise_max(__builtin_bit_cast(s16x8_fix, a), __builtin_bit_cast(s16x8_fix, b)));
}
__global__ __launch_bounds__(256) void pool_seam_fix_kernel(PoolFixParams p) {
    const int nbr = p.tiles_h - 1, nbc = p.tiles_w - 1, PT = p.PT;
    const int V = p.C / 8;
    const long long per_plane = ((long long)nbr * p.Wp + (long long)nbc * p.Hp) * V;
    const long long total = per_plane * p.planes;
    unsigned short* yg = (unsigned short*)p.y;
    const unsigned short* rb = (const unsigned short*)p.rowbuf;
    const unsigned short* cb = (const unsigned short*)p.colbuf;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)blockDim.x * gridDim.x) {
        const long long plane = idx / per_plane;
        long long it = idx % per_plane;
        const int v = (int)(it % V);
        it /= V;
        int ph, pw;
        bool row_item;
        if (it < (long long)nbr * p.Wp) { row_item = true; ph = PT * ((int)(it / p.Wp) + 1) - 1; pw = (int)(it % p.Wp); }
        else { it -= (long long)nbr * p.Wp; row_item = false; pw = PT * ((int)(it / p.Hp) + 1) - 1; ph = (int)(it % p.Hp); }
        if (ph >= p.Hp || pw >= p.Wp) continue;
        const bool ph_seam = ((ph + 1) % PT) == 0 && ((ph + 1) / PT) <= nbr;
        const bool pw_seam = ((pw + 1) % PT) == 0 && ((pw + 1) / PT) <= nbc;
        if (!row_item && ph_seam) continue;                       // a corner: the row pass owns it
        unsigned short* yp = yg + ((plane * p.Hp + ph) * p.Wp + pw) * (size_t)p.y_cstride + p.y_coff + v * 8;
        u32x4 m = *(const u32x4*)yp;
        if (row_item) {
            const int t = (ph + 1) / PT;
            for (int dc = 0; dc < 3; ++dc) {
                const int c = 2 * pw + dc;
                if (c < p.W) m = pk_max_nonneg16_fix(m, *(const u32x4*)(rb + ((plane * p.tiles_h + t) * p.W + c) * (size_t)p.C + v * 8));
            }
            if (pw_seam) {
                const int s_ = (pw + 1) / PT;
                for (int dr = 0; dr < 2; ++dr)
                    m = pk_max_nonneg16_fix(m, *(const u32x4*)(cb + ((plane * p.tiles_w + s_) * p.H + 2 * ph + dr) * (size_t)p.C + v * 8));
            }
        } else {
            const int s_ = (pw + 1) / PT;
            for (int dr = 0; dr < 3; ++dr) {
                const int r = 2 * ph + dr;
                if (r < p.H) m = pk_max_nonneg16_fix(m, *(const u32x4*)(cb + ((plane * p.tiles_w + s_) * p.H + r) * (size_t)p.C + v * 8));
            }
        }
        *(u32x4*)yp = m;
    }
}


}  // namespace step
extern "C" {

// supported: what step_conv_forward_pre takes, planned onto the 4-plane 8x8 tile (maps whose sides the planner tiles by 8: 56 x 56 at C2),
// ReLU on (the pooled epilogue orders 16-bit patterns as integers: values must be >= +0), 16-byte output vectors, no residual
static bool conv_pre_pool_plan(const step_conv_desc* d, step_conv_desc& canon, ConvPlan& pl, bool& p2) {
    if (!d || d->N <= 0 || d->D <= 0 || d->H <= 0 || d->W <= 0) return false;
    canon = canonical_desc(d);
    if (canon.Cin != 64 || canon.dtype == STEP_F32 || canon.split || !(canon.kd == 3 && canon.kh == 3 && canon.kw == 3) || !canon.relu) return false;
    if (canon.Cout % 8 || canon.y_cstride % 8 || canon.y_coff % 8) return false;
    if ((unsigned long long)canon.N * canon.D * canon.H * canon.W * (unsigned long long)canon.x_cstride >= 0xffffffffULL) return false;
    pl = conv_plan(&canon);
    p2 = false;
    if (pl.ok && pl.impl == 1 && pl.ph == 1 && pl.wv == 8 && pl.tps == 2 && pl.twl == 3) return true;
    // The planner prefers a general box (e.g. the 100 x 100 x 18-plane maps of AVA clips: 720 boxes of 2 x 5 x 25 against 845 tiles of
    // 4 x 8 x 8).  With the pool riding in the epilogue the 4 x 8 x 8 tiling wins all the same: measured at that shape, 4 clips, bf16
    // (tools/prepool_ab.py): general box + stand-alone pool 570.6 us, 4 x 8 x 8 + stand-alone pool 599.1, 4 x 8 x 8 fused 521.2.  Taken
    // when it needs at most 25 % more tiles than the planner's own choice.
    const ConvPlan q = conv_plan(&canon, true, 0, true);
    if (pl.ok && q.ok && q.impl == 1 && q.ph == 1 && q.wv == 8 && q.tps == 2 && q.twl == 3 && q.mtiles * 4 <= pl.mtiles * 5) { pl = q; p2 = true; return true; }
    return false;
}

size_t step_conv_pre_pool_workspace_bytes(const step_conv_desc* d) {
    step_conv_desc canon;
    ConvPlan pl;
    bool p2;
    if (!conv_pre_pool_plan(d, canon, pl, p2)) return 0;
    const size_t planes = (size_t)canon.N * canon.D;
    return planes * ((size_t)pl.tiles_h * canon.W + (size_t)pl.tiles_w * canon.H) * canon.Cout * 2;
}

// parts: 1 = the conv launches (tiles + their first rows / columns), 2 = the seam pass, 3 = both
static int conv_forward_pre_pool_impl(const step_conv_desc* d, const void* x, const void* w_packed, const float* scale, const float* shift,
                                      const void* pre_w_packed, const float* pre_scale, const float* pre_shift, int pre_cin, void* y_pooled,
                                      void* ws, size_t ws_bytes, int parts, step_stream_t stream) {
    step_conv_desc canon;
    ConvParams p;
    const bool conv = (parts & 1) != 0;
    // (the seam pass needs the descriptor, the pooled tensor and the workspace only)
    const int rc = conv_fill_params(d, conv ? x : y_pooled, conv ? w_packed : y_pooled, scale, shift, nullptr, y_pooled, nullptr, canon, p);
    if (rc != STEP_OK) return rc;
    if (conv && !pre_w_packed) return STEP_E_NULL;
    if (p.N == 0) return STEP_OK;
    ConvPlan pl;
    bool p2 = false;
    if ((conv && pre_cin != 64) || !conv_pre_pool_plan(d, canon, pl, p2) || !p.vec_epi) return STEP_E_UNSUPPORTED;
    p.pool_p2 = p2 ? 1 : 0;
    if (conv && (((uintptr_t)pre_w_packed % 16) || (pre_scale && ((uintptr_t)pre_scale % 16)) || (pre_shift && ((uintptr_t)pre_shift % 16)))) return STEP_E_ALIGN;
    const size_t need = step_conv_pre_pool_workspace_bytes(d);
    if (!ws || ws_bytes < need || ((uintptr_t)ws % 16)) return STEP_E_SHAPE;
    p.pre_w = pre_w_packed; p.pre_scale = pre_scale; p.pre_shift = pre_shift;
    const size_t planes = (size_t)canon.N * canon.D;
    p.pool_row = ws;
    p.pool_col = (unsigned char*)ws + planes * (size_t)pl.tiles_h * canon.W * canon.Cout * 2;
    p.Hp = step_pool_out_size(canon.H, 3, 2); p.Wp = step_pool_out_size(canon.W, 3, 2);      // (1,3,3) / (1,2,2), TF padding (0,1), ceil mode
    if (conv) {
        const int rc2 = canon.dtype == STEP_BF16 ? conv_forward_t<bf16_t>(&canon, p, nullptr, 0, stream) : conv_forward_t<f16_t>(&canon, p, nullptr, 0, stream);
        if (rc2 != STEP_OK) return rc2;
    }
    if ((parts & 2) && (pl.tiles_h > 1 || pl.tiles_w > 1)) {
        PoolFixParams f;
        f.y = y_pooled; f.rowbuf = p.pool_row; f.colbuf = p.pool_col; f.planes = (long long)planes;
        f.H = canon.H; f.W = canon.W; f.Hp = p.Hp; f.Wp = p.Wp; f.tiles_h = pl.tiles_h; f.tiles_w = pl.tiles_w; f.C = canon.Cout;
        f.y_cstride = canon.y_cstride; f.y_coff = canon.y_coff; f.PT = 4;
        const long long items = (long long)planes * ((long long)(pl.tiles_h - 1) * p.Wp + (long long)(pl.tiles_w - 1) * p.Hp) * (canon.Cout / 8);
        if (items > 0) {
            STEP_LAUNCH(pool_seam_fix_kernel, dim3(flat_grid(items, 256)), dim3(256), stream, f);
            return STEP_LAUNCH_CHECK();
        }
    }
    return STEP_OK;
}

int step_conv_forward_pre_pool(const step_conv_desc* d, const void* x, const void* w_packed, const float* scale, const float* shift,
                               const void* pre_w_packed, const float* pre_scale, const float* pre_shift, int pre_cin, void* y_pooled,
                               void* ws, size_t ws_bytes, step_stream_t stream) {
    return conv_forward_pre_pool_impl(d, x, w_packed, scale, shift, pre_w_packed, pre_scale, pre_shift, pre_cin, y_pooled, ws, ws_bytes, 3, stream);
}

int step_conv_forward_pre_pool_tiles(const step_conv_desc* d, const void* x, const void* w_packed, const float* scale, const float* shift,
                                     const void* pre_w_packed, const float* pre_scale, const float* pre_shift, int pre_cin, void* y_pooled,
                                     void* ws, size_t ws_bytes, step_stream_t stream) {
    return conv_forward_pre_pool_impl(d, x, w_packed, scale, shift, pre_w_packed, pre_scale, pre_shift, pre_cin, y_pooled, ws, ws_bytes, 1, stream);
}

int step_conv_pre_pool_finish(const step_conv_desc* d, void* y_pooled, void* ws, size_t ws_bytes, step_stream_t stream) {
    return conv_forward_pre_pool_impl(d, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 64, y_pooled, ws, ws_bytes, 2, stream);
}

// Can these convs share one grid?  All 16-bit 3x3x3 layers the planner sends to the two-phase conv_tap form on general boxes
// (forced to 8 waves: a member that alone would take the four-wave form for its small grid is exactly what a group is for).
static bool conv_group_plan(const step_conv_item* items, int n, step_conv_desc* canon, ConvParams* ps, ConvPlan* pls, int* NBc) {
    if (n < 2 || n > CONV_GROUP_MAX) return false;
    if (opt(STEP_OPT_CONV_IMPL) != -1 || opt(STEP_OPT_CONV_NB) != 0 || opt(STEP_OPT_CONV_WAVES) == 4 || opt(STEP_OPT_CONV_PHASED) == 0) return false;
    int nb = 0;
    for (int k = 0; k < n; ++k) {
        const step_conv_desc* d = &canon[k];
        if (d->dtype == STEP_F32 || d->dtype != canon[0].dtype) return false;
        if (!(d->kd == 3 && d->kh == 3 && d->kw == 3) || ps[k].N == 0) return false;
        constexpr int VEC = 8;
        if (d->Cin % VEC || d->x_cstride % VEC || d->x_coff % VEC) return false;
        if (((uintptr_t)ps[k].x % 16) || ((uintptr_t)ps[k].w % 16)) return false;
        pls[k] = conv_plan(d, true, 8);
        const ConvPlan& pl = pls[k];
        if (!pl.ok || pl.impl != 1 || pl.ph != 1 || pl.wv != 8 || pl.tps != 2 || (pl.twl != 0 && pl.twl != 3) || pl.twl != pls[0].twl) return false;
        if (pl.NB > nb) nb = pl.NB;
    }
    *NBc = nb;
    return true;
}

// The members of a grouped launch: the 3x3x3 items that share one conv_tap instantiation and, optionally, ONE pointwise item the
// planner sends to conv_pw_kernel<T, 1, 4> (plain epilogue).  tap[] / pw index into items; returns false when the items do not
// group (the caller launches them one by one).
struct GroupSel { int ntap, tap[CONV_GROUP_MAX], pw; };
static bool conv_group_select(const step_conv_item* items, int n, GroupSel& sel) {
    sel.ntap = 0; sel.pw = -1;
    if (n < 2 || n > CONV_GROUP_MAX + 1) return false;
    for (int k = 0; k < n; ++k) {
        const step_conv_desc* d = items[k].desc;
        if (!d) return false;
        if (d->kd == 1 && d->kh == 1 && d->kw == 1) {
            if (sel.pw >= 0) return false;
            sel.pw = k;
        } else {
            if (sel.ntap == CONV_GROUP_MAX) return false;
            sel.tap[sel.ntap++] = k;
        }
    }
    return sel.ntap == CONV_GROUP_MAX || (sel.ntap >= 2 && sel.pw < 0);
}

// the pointwise member's parameter block, or false when the layer is not one the group kernel carries
static bool conv_group_pw_params(const step_conv_item& it, int dtype, step_conv_desc& canon, ConvParams& p, long long base) {
    if (conv_fill_params(it.desc, it.x, it.w_packed, it.scale, it.shift, it.res, it.y, nullptr, canon, p) != STEP_OK || p.N == 0) return false;
    if (canon.dtype != dtype || it.desc->split || it.res) return false;
    constexpr int VEC = 8;
    if (canon.Cin % VEC || canon.x_cstride % VEC || canon.x_coff % VEC || ((uintptr_t)p.x % 16) || ((uintptr_t)p.w % 16)) return false;
    const ConvPlan pl = conv_plan(&canon, false);
    if (!pl.ok || pl.impl != 2 || pl.NB != 1 || pl.wv != 4) return false;
    p.tiles_h = pl.tiles_h; p.tiles_w = pl.tiles_w; p.tiles_d = pl.tiles_d;
    p.gtd = pl.gtd; p.gth = pl.gth; p.gtw = pl.gtw; p.gmode = pl.gmode;
    const int groups = ceil_div(p.nblk32, 2);
    p.gx = (int)pl.mtiles; p.gy = groups;
    const long long tot = (pl.mtiles * groups + 7) / 8 * 8;
    if (base + tot > 0x7fffffffLL) return false;
    p.gbase = (int)base; p.gcount = (int)tot;
    return true;
}

int step_conv_forward_group(const step_conv_item* items, int n, step_stream_t stream) {
    if (n < 0) return STEP_E_SHAPE;
    if (n == 0) return STEP_OK;
    if (!items) return STEP_E_NULL;
    GroupSel sel;
    bool done[CONV_GROUP_MAX + 1] = {false, false, false};
    if (conv_group_select(items, n, sel)) {
        step_conv_desc canon[CONV_GROUP_MAX];
        ConvParams ps[CONV_GROUP_MAX];
        ConvPlan pls[CONV_GROUP_MAX];
        step_conv_item taps[CONV_GROUP_MAX];
        bool ok = true;
        for (int k = 0; k < sel.ntap && ok; ++k) {
            taps[k] = items[sel.tap[k]];
            const step_conv_item& it = taps[k];
            const int rc = conv_fill_params(it.desc, it.x, it.w_packed, it.scale, it.shift, it.res, it.y, nullptr, canon[k], ps[k]);
            if (rc != STEP_OK) return rc;
            ok = ps[k].N != 0 && (!it.desc->split);
        }
        int NBc = 0;
        if (ok && conv_group_plan(taps, sel.ntap, canon, ps, pls, &NBc)) {
            ConvGroupParams g;
            g.n = sel.ntap;
            // longest workgroups first (they are dispatched first): descending K depth
            int order[CONV_GROUP_MAX];
            for (int k = 0; k < sel.ntap; ++k) order[k] = k;
            if (sel.ntap == 2 && (long long)canon[1].Cin * pls[1].NB > (long long)canon[0].Cin * pls[0].NB) { order[0] = 1; order[1] = 0; }
            long long base = 0;
            for (int j = 0; j < sel.ntap; ++j) {
                const int k = order[j];
                ConvParams& p = g.p[j];
                p = ps[k];
                const ConvPlan& pl = pls[k];
                p.tiles_h = pl.tiles_h; p.tiles_w = pl.tiles_w; p.tiles_d = pl.tiles_d;
                p.gtd = pl.gtd; p.gth = pl.gth; p.gtw = pl.gtw; p.gmode = pl.gmode;
                conv_set_box_magic(p, canon[k].kh, canon[k].kw);
                const int groups = ceil_div(p.nblk32, 2 * NBc);
                p.gx = (int)pl.mtiles; p.gy = groups;
                const long long tot = (pl.mtiles * groups + 7) / 8 * 8;
                p.gbase = (int)base; p.gcount = (int)tot;
                base += tot;
            }
            for (int j = sel.ntap; j < CONV_GROUP_MAX; ++j) g.p[j] = g.p[0];
            g.pw = g.p[0];
            g.pw.gbase = 0; g.pw.gcount = 0;                 // (gcount == 0: no pointwise member)
            // the pointwise member rides along when the 3x3x3 members leave CUs idle (fewer one-per-CU workgroups than CUs):
            // behind a launch that fills every CU its workgroups would only queue
            bool with_pw = false;
            if (sel.pw >= 0 && pls[0].twl == 0 && base <= opt(STEP_OPT_CONV_GROUP_PW) && opt(STEP_OPT_THROUGHPUT) == 0) {
                step_conv_desc cpw;
                with_pw = conv_group_pw_params(items[sel.pw], canon[0].dtype, cpw, g.pw, base);
                if (!with_pw) { g.pw = g.p[0]; g.pw.gbase = 0; g.pw.gcount = 0; }
            }
            const long long total = base + (with_pw ? g.pw.gcount : 0);
            if (total <= 0x7fffffffLL) {
                const dim3 grid((unsigned)total);
                const int rc = canon[0].dtype == STEP_BF16 ? conv_tap_group_launch<bf16_t>(pls[0].twl, NBc, g, grid, stream)
                                                           : conv_tap_group_launch<f16_t>(pls[0].twl, NBc, g, grid, stream);
                if (rc != STEP_OK) return rc;
                for (int k = 0; k < sel.ntap; ++k) done[sel.tap[k]] = true;
                if (with_pw) done[sel.pw] = true;
            }
        }
    }
    for (int k = 0; k < n; ++k) {                            // what did not group: one launch each (the same results)
        if (k <= CONV_GROUP_MAX && done[k]) continue;
        const step_conv_item& it = items[k];
        const int rc = step_conv_forward_ws(it.desc, it.x, it.w_packed, it.scale, it.shift, it.res, it.y, nullptr, nullptr, 0, stream);
        if (rc != STEP_OK) return rc;
    }
    return STEP_OK;
}

int step_conv_group_kernel_name(const step_conv_item* items, int n, char* buf, int buflen) {
    if (!items || !buf || buflen <= 0) return STEP_E_NULL;
    buf[0] = 0;
    GroupSel sel;
    if (!conv_group_select(items, n, sel)) return STEP_OK;
    step_conv_desc canon[CONV_GROUP_MAX];
    ConvParams ps[CONV_GROUP_MAX];
    ConvPlan pls[CONV_GROUP_MAX];
    step_conv_item taps[CONV_GROUP_MAX];
    for (int k = 0; k < sel.ntap; ++k) {
        taps[k] = items[sel.tap[k]];
        const step_conv_item& it = taps[k];
        if (conv_fill_params(it.desc, it.x, it.w_packed, it.scale, it.shift, it.res, it.y, nullptr, canon[k], ps[k]) != STEP_OK || ps[k].N == 0 || it.desc->split)
            return STEP_OK;
    }
    int NBc = 0;
    if (!conv_group_plan(taps, sel.ntap, canon, ps, pls, &NBc)) return STEP_OK;
    long long base = 0;
    for (int k = 0; k < sel.ntap; ++k) base += (pls[k].mtiles * ceil_div(ps[k].nblk32, 2 * NBc) + 7) / 8 * 8;
    bool with_pw = false;
    if (sel.pw >= 0 && pls[0].twl == 0 && base <= opt(STEP_OPT_CONV_GROUP_PW) && opt(STEP_OPT_THROUGHPUT) == 0) {
        step_conv_desc cpw;
        ConvParams ppw;
        with_pw = conv_group_pw_params(items[sel.pw], canon[0].dtype, cpw, ppw, base);
    }
    const char* t = canon[0].dtype == STEP_BF16 ? "step::bf16_t" : "step::f16_t";
    snprintf(buf, (size_t)buflen, "void step::conv_tap_group%s_kernel<%s, %d, %d, 3, 3, 3, 2, 2, 8, 1>(step::ConvGroupParams)", with_pw ? "_pw" : "", t,
             pls[0].twl, NBc);
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
    else if (pl.impl == 4) {
        int nb, ksteps;
        pws_shape(pl.NB, ceil_div(d->Cin, CK) * 2, nb, ksteps);
        const long long M_ = (long long)d->N * d->D * d->H * d->W;
        const bool w16 = d->res_cstride == 0 && pws_sixteen(M_, pl.mtiles);
        if (d->res_cstride != 0) {
            snprintf(buf, (size_t)buflen, "void step::conv_pws_kernel<%s, %d, %d, 8, true>(step::ConvParams, int)", t, nb, ksteps);
        } else snprintf(buf, (size_t)buflen, "void step::conv_pws_kernel<%s, %d, %d, %d, false>(step::ConvParams, int)", t, w16 ? (nb > 2 ? 2 : nb) : nb, ksteps, w16 ? 16 : 8);
    }
    else if (pl.impl == 2)
        snprintf(buf, (size_t)buflen, "void step::conv_pw_kernel<%s, %d, %d>(step::ConvParams)", t, pl.NB, pl.wv);
    else if (pl.impl == 1)
        snprintf(buf, (size_t)buflen, "void step::conv_tap_kernel<%s, %d, %d, %d, %d, %d, %d, %d, %d, %d>(step::ConvParams)", t,
                 pl.twl, pl.NB, d->kd, d->kh, d->kw, pl.tps, pl.mb, pl.wv, pl.wv == 8 ? pl.ph : 0);
    else
        snprintf(buf, (size_t)buflen, "void step::conv_igemm_kernel<%s, %d, %d, %d, %d, %d, %s, %d>(step::ConvParams)", t,
                 pl.flat ? 4 : (pl.wide ? 5 : 4), pl.NB, d->kd, d->kh, d->kw, pl.flat ? "true" : "false", pl.deep ? 128 : 32);
    return STEP_OK;
}

int step_conv_plan_info(const step_conv_desc* d, int* info, int n) {
    if (!d || !info || n < 10) return STEP_E_NULL;
    const step_conv_desc canon = canonical_desc(d);
    const ConvPlan pl = conv_plan(&canon);
    if (!pl.ok) return STEP_E_UNSUPPORTED;
    info[0] = pl.impl; info[1] = pl.twl; info[2] = pl.NB; info[3] = pl.wv; info[4] = pl.ph; info[5] = pl.gtd; info[6] = pl.gth; info[7] = pl.gtw;
    info[8] = pl.gmode; info[9] = (int)(pl.mtiles > 0x7fffffff ? 0x7fffffff : pl.mtiles);
    return STEP_OK;
}

int step_conv_pre_pool_plan_info(const step_conv_desc* d, int* info, int n) {
    if (!d || !info || n < 12) return STEP_E_NULL;
    step_conv_desc canon;
    ConvPlan pl;
    bool p2 = false;
    if (!conv_pre_pool_plan(d, canon, pl, p2)) return STEP_E_UNSUPPORTED;
    info[0] = pl.impl; info[1] = pl.twl; info[2] = pl.NB; info[3] = pl.wv; info[4] = pl.ph; info[5] = pl.gtd; info[6] = pl.gth; info[7] = pl.gtw;
    info[8] = pl.gmode; info[9] = (int)(pl.mtiles > 0x7fffffff ? 0x7fffffff : pl.mtiles);
    info[10] = pl.tiles_h; info[11] = pl.tiles_w;
    if (n >= 13) {                                         // info[12]: workgroups of the persistent tile loop the NB = 3 launch runs as (0: one workgroup per tile)
        const long long slots_ = opt(STEP_OPT_CONV_SLOTS) > 0 ? opt(STEP_OPT_CONV_SLOTS) : 256;
        long long main_tiles = pl.mtiles;
        const long long tail = pl.mtiles % slots_;
        const int tgroups = ceil_div(ceil_div(canon.Cout, 32), 2);
        if (opt(STEP_OPT_CONV_TAIL) != 0 && pl.NB > 1 && pl.mtiles > slots_ && tail > 0 && tail * 4 <= slots_ && tail * tgroups <= slots_) main_tiles -= tail;
        const bool one_group = ceil_div(ceil_div(canon.Cout, 32), 2 * pl.NB) == 1;
        info[12] = (opt(STEP_OPT_CONV_PERSIST) != 0 && one_group && pl.NB == 3 && main_tiles > slots_) ? (int)(slots_ / 8 * 8 > 0 ? slots_ / 8 * 8 : 8) : 0;
    }
    return STEP_OK;
}

#ifdef STEP_PROBE
__attribute__((visibility("default"))) void step_probe_set(void* buf) { step::g_probe_buf = (unsigned long long*)buf; }
#endif
const char* step_version(void) { return "step_amd 0.1.0 gfx950"; }
int step_abi_version(void) { return 35; }

}  // extern "C"

