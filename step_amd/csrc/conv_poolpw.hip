// step_amd/csrc/conv_poolpw.hip -- the branch_3 path of an Inception block as ONE kernel:
//     y = act( conv1x1x1( maxpool3d_tf(x; 3x3x3, stride 1) ) * scale[c] + shift[c] )
// replaces  MaxPool3dTFPadding((3,3,3),(1,1,1)) -> Unit3Dpy(1x1x1)  (models/i3dpt.py:151-155, 160) -- in the reference a pad
// copy, a pool pass that writes all Cin channels of the block input again and a conv pass that reads them back.  Here the
// pooled tensor never exists in memory: 7 pool launches and ~0.4 GB of HBM traffic per C2 step disappear.
//
// 256 threads = 4 wavefronts own a box of TD x TH x TW <= 128 output pixels (chosen at launch) x all (<= 128) output
// channels of a channel group and walk the input channels in 64-byte slabs (32 x 16-bit / 16 x fp32):
//   stage : the slab of the (TD+2) x (TH+2) x (TW+2) input halo goes global -> registers -> LDS (64 B per pixel, linear);
//           a position outside the image holds the VALUE 0 -- the reference's explicit zero pad, which takes part in the
//           max -- and the next slab's loads are issued before this slab's pool pass, so they fly under it;
//   pool  : a thread owns an (h, w, 16-byte channel vector) column of the box: per input plane 9 LDS reads give the 2-D window
//           max, a rolling 3-plane max along D gives the pooled vector (the 2-D maxima are shared by up to three output
//           planes), written into the A tile in the pitch-80 pixel-major form the MFMA fragment reads want.  All comparisons
//           run in the storage type (VecMax: packed 16-bit integer max on re-keyed bf16);
//   gemm  : wave w multiplies 32 pixels of the A tile with the slab's weights -- B fragments straight from the packed
//           weights in global memory (a few KiB per slab, L2-resident, requested before the pool pass) -- into its fp32
//           accumulators.
// S2 = true: the same with the (1,3,3) / (1,2,2) pool of maxPool3d_2a_3x3 in front of conv3d_2b_1x1 (models/i3dpt.py:193-201):
// planes are independent, the halo of a TD x TH x TW output box is TD x (2 TH + 1) x (2 TW + 1) input pixels, the TF pad is
// back-heavy (window of output (h, w) = input rows 2h .. 2h+2: no front pad), every (plane, h, w, vector) is its own pool column.
// Two barriers per slab; with ~45 KiB of LDS three workgroups share a CU, so one workgroup's pool pass (LDS-bound) runs under
// another's loads.  WN = 2 (boxes of <= 64 pixels, e.g. one 7 x 7 quadrant plane of a 14 x 14 map): the four waves are
// 2 (pixels) x 2 (channel halves) instead of 4 x 1.
#include "conv_common.h"
#include "pool_vec.h"

namespace step {

constexpr int PPW_NPIX = 512;            // halo pixels of a box (32 KiB of LDS)
constexpr int PPW_PK = 2;                // pool columns per thread: TH*TW*4 <= 512

// DEPTH = slabs of the halo in flight (register sets): 1 measured latency-bound (3.2 us per slab against 0.4 us of work);
// with DEPTH sets the loads of slab s + DEPTH are issued while slab s is processed.  The slab loop is unrolled by DEPTH and
// branch-free: slabs past the end stage zeros (a zero pooled tile against finite weights adds nothing).
template <typename T, int NBW, int WN, bool S2 = false, int DEPTH = 3>
__global__ __launch_bounds__(256, 2) void pool_pw_kernel(ConvParams p) {
    static_assert(WN == 1 || WN == 2, "4 x 1 or 2 x 2 waves");
    constexpr int NT = 256;
    constexpr int WM = 4 / WN;                  // waves along the pixel axis
    constexpr int TPXM = WM * 32;               // pixel capacity of the box: 128 or 64
    constexpr int NBT = NBW * WN;               // 32-channel output blocks per workgroup
    constexpr int ES = (int)sizeof(T);
    constexpr int VEC = 16 / ES;
    constexpr int CKT = 64 / ES, KS = CKT / 16;
    constexpr int HP = 64, AP = 80;             // bytes per halo pixel / per A-tile pixel
    constexpr int ITER = PPW_NPIX * 4 / NT;
    typedef typename Ld16<T>::type raw;         // 16-byte channel vector in the storage type
    typedef typename frag<T>::type frag_t;
    constexpr int OTP = 36;                     // floats per row of the epilogue's per-wave transpose tile
    constexpr int HALO_B = PPW_NPIX * HP, A_B = 128 * AP;
    static_assert(HALO_B >= 4 * 32 * OTP * 4 + 512, "the epilogue reuses the halo region");
    __shared__ __attribute__((aligned(16))) unsigned char lds[HALO_B + A_B];
    unsigned char* const halo = lds;
    unsigned char* const ldsA = lds + HALO_B;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
#ifdef STEP_EMUL
    const int wave = tid >> 6;
#else
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
    const int khalf = lane >> 5;
    const int wm = wave % WM, wn = wave / WM;

    int gbx, gby;
    if (!grid_coords(p, gbx, gby)) return;
    const int TD = p.gtd, TH = p.gth, TW = p.gtw;
    const int TPX = TD * TH * TW;
    const int HH_ = S2 ? 2 * TH + 1 : TH + 2, HW_ = S2 ? 2 * TW + 1 : TW + 2, PD = S2 ? TD : TD + 2;
    const int Hin = S2 ? p.Hi : p.H, Win = S2 ? p.Wi : p.W;      // input extent (S2: the pool halves H and W)
    const int HHW = HH_ * HW_;
    const int NVEC = PD * HHW * 4;
    int t = gbx;
    const int tw_i = t % p.tiles_w; t /= p.tiles_w;
    const int th_i = t % p.tiles_h; t /= p.tiles_h;
    const int d0 = (t % p.tiles_d) * TD;
    const int n = t / p.tiles_d;
    const int h0 = th_i * TH, w0 = tw_i * TW;
    const int nb0 = gby * NBT;
    const int KC16 = p.nchunks32 * 2;
    const int nslab = (p.Cin + CKT - 1) / CKT;

    const unsigned char* xg = (const unsigned char*)p.x;
    const T* wg = (const T*)p.w;

    // ---- staging table: element offset of the pixel behind each of this thread's halo vectors (slot = tid & 3 for all of them)
    const int slotc = (tid & 3) * VEC;
    unsigned goff[ITER], gmask[ITER];
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int v = tid + it * NT;
        goff[it] = 0u; gmask[it] = 0u;
        if (v < NVEC) {
            const int pix = v >> 2;
            const int plane = pix / HHW, rem = pix % HHW;
            const int r = rem / HW_, cc = rem % HW_;
            const int id = S2 ? d0 + plane : d0 + plane - 1, ih = S2 ? 2 * h0 + r : h0 + r - 1, iw = S2 ? 2 * w0 + cc : w0 + cc - 1;
            if (id >= 0 && id < p.D && ih >= 0 && ih < Hin && iw >= 0 && iw < Win) {
                const size_t gpix = (((size_t)n * p.D + id) * Hin + ih) * Win + iw;
                goff[it] = (unsigned)(gpix * p.x_cstride + p.x_coff);
                gmask[it] = 0xffffffffu;
            }
        }
    }
    u32x4 stage[DEPTH][ITER];
    // branch-free loads (a predicated load makes the compiler drain vmcnt): positions outside the image re-read pixel 0 of
    // the tensor, channels past Cin re-read channel 0; both are masked when the vector is written to LDS
    auto load_slab = [&](int slab, auto setc) {
        constexpr int SET = decltype(setc)::value;
        const int c = slab * CKT + slotc;
        const int ce = (slab < nslab && c < p.Cin) ? c : 0;
#pragma unroll
        for (int it = 0; it < ITER; ++it) stage[SET][it] = *(const u32x4*)(xg + ((size_t)goff[it] + ce) * ES);
    };
    auto store_slab = [&](int slab, auto setc) {
        constexpr int SET = decltype(setc)::value;
        const unsigned cm = (slab * CKT + slotc < p.Cin) ? 0xffffffffu : 0u;
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int v = tid + it * NT;
            if (v < NVEC) {
                const u32x4 mv = stage[SET][it] & (gmask[it] & cm);
                *(raw*)(halo + v * 16) = VecMax<T>::enc(__builtin_bit_cast(raw, mv));
            }
        }
    };

    // ---- pool columns of this thread: (h, w, slot) -> LDS offsets of the window's first vector / of the A-tile row
    int hoff[PPW_PK], aoff[PPW_PK];
    const int ncol = (S2 ? TD : 1) * TH * TW * 4;              // S2: every plane of the box is its own set of columns
#pragma unroll
    for (int k = 0; k < PPW_PK; ++k) {
        const int item = tid + k * NT;
        const int hw = item >> 2, slot = item & 3;
        if (S2) {
            const int w = hw % TW, q = hw / TW;
            const int h = q % TH, d = q / TH;
            hoff[k] = item < ncol ? ((d * HH_ + 2 * h) * HW_ + 2 * w) * HP + slot * 16 : -1;
            aoff[k] = hw * AP + slot * 16;                      // tile pixel index = (d * TH + h) * TW + w = hw
        } else {
            const int h = hw / TW, w = hw % TW;
            hoff[k] = item < ncol ? (h * HW_ + w) * HP + slot * 16 : -1;
            aoff[k] = (h * TW + w) * AP + slot * 16;
        }
    }
    const int planeB = HHW * HP, rowB = HW_ * HP, aplaneB = TH * TW * AP;
    const raw klow = VecMax<T>::lowest();
    auto pool_pass = [&]() {
#pragma unroll
        for (int k = 0; k < PPW_PK; ++k) {
            if (hoff[k] < 0) continue;
            raw m1 = klow, m2 = klow;
            const unsigned char* b = halo + hoff[k];
            unsigned char* a = ldsA + aoff[k];
            if (S2) {                                        // one plane, one 3 x 3 window
                raw m = *(const raw*)b;
                m = VecMax<T>::max(m, *(const raw*)(b + HP));
                m = VecMax<T>::max(m, *(const raw*)(b + 2 * HP));
                m = VecMax<T>::max(m, *(const raw*)(b + rowB));
                m = VecMax<T>::max(m, *(const raw*)(b + rowB + HP));
                m = VecMax<T>::max(m, *(const raw*)(b + rowB + 2 * HP));
                m = VecMax<T>::max(m, *(const raw*)(b + 2 * rowB));
                m = VecMax<T>::max(m, *(const raw*)(b + 2 * rowB + HP));
                m = VecMax<T>::max(m, *(const raw*)(b + 2 * rowB + 2 * HP));
                *(raw*)a = VecMax<T>::dec(m);
                continue;
            }
            for (int pd = 0; pd < PD; ++pd) {
                raw m = *(const raw*)b;
                m = VecMax<T>::max(m, *(const raw*)(b + HP));
                m = VecMax<T>::max(m, *(const raw*)(b + 2 * HP));
                m = VecMax<T>::max(m, *(const raw*)(b + rowB));
                m = VecMax<T>::max(m, *(const raw*)(b + rowB + HP));
                m = VecMax<T>::max(m, *(const raw*)(b + rowB + 2 * HP));
                m = VecMax<T>::max(m, *(const raw*)(b + 2 * rowB));
                m = VecMax<T>::max(m, *(const raw*)(b + 2 * rowB + HP));
                m = VecMax<T>::max(m, *(const raw*)(b + 2 * rowB + 2 * HP));
                if (pd >= 2) {
                    *(raw*)a = VecMax<T>::dec(VecMax<T>::max(m, VecMax<T>::max(m1, m2)));
                    a += aplaneB;
                }
                m2 = m1; m1 = m;
                b += planeB;
            }
        }
    };

    // ---- GEMM operands
    const unsigned char* const abase = ldsA + (wm * 32 + (lane & 31)) * AP + khalf * (ES == 4 ? 32 : 16);
    const T* wthr[NBW];
#pragma unroll
    for (int i = 0; i < NBW; ++i) {
        const int nbg = min(nb0 + wn * NBW + i, p.nblk32 - 1);      // blocks past Cout compute on a duplicate and are never stored
        wthr[i] = wg + ((size_t)nbg * KC16 * 64 + lane) * 8;
    }
    f32x16 acc[NBW];
#pragma unroll
    for (int i = 0; i < NBW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    static_for<DEPTH>([&](auto sc) { load_slab(decltype(sc)::value, sc); });
    auto process = [&](int s, auto setc) {
        store_slab(s, setc);                              // (slabs past the end: zeros)
        __syncthreads();                                  // halo slab visible; every wave is done with the previous A tile
        frag_t fb[NBW][KS];
        const int sb = min(s, nslab - 1);                 // (past the end: finite weights against a zero tile)
#pragma unroll
        for (int i = 0; i < NBW; ++i)
#pragma unroll
            for (int j = 0; j < KS; ++j) fb[i][j] = load_b_frag<T>(wthr[i] + (size_t)(sb * KS + j) * 512);
        load_slab(s + DEPTH, setc);                       // (past the end: a harmless re-read)
        pool_pass();
        __syncthreads();                                  // A tile complete; the halo may be overwritten
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            const frag_t fa = lds_read_bfrag<T>(abase + j * 32);
#pragma unroll
            for (int i = 0; i < NBW; ++i) mma_k16(fa, fb[i][j], acc[i], T());
        }
    };
#pragma unroll 1
    for (int s0 = 0; s0 < nslab; s0 += DEPTH) static_for<DEPTH>([&](auto sc) { process(s0 + decltype(sc)::value, sc); });

    // ---- epilogue: affine + ReLU, channels-last store
    T* yg = (T*)p.y;
    int* const pixtab = (int*)(lds + 4 * 32 * OTP * 4);
    __syncthreads();                                      // (the last A-tile reads are done before the region is reused)
    if (tid < TPXM) {
        const int mc = tid < TPX ? tid : 0;
        const int twl = mc % TW, q = mc / TW;
        const int thl = q % TH, tdl = q / TH;
        const int od = d0 + tdl, oh = h0 + thl, ow = w0 + twl;
        const bool ok = tid < TPX && od < p.D && oh < p.H && ow < p.W;
        pixtab[tid] = ok ? (int)((((long long)n * p.D + od) * p.H + oh) * p.W + ow) : -1;
    }
    if (ES == 2 && p.vec_epi) {
        float* const ot = (float*)lds + wave * 32 * OTP;
#pragma unroll
        for (int i = 0; i < NBW; ++i) {
            const int cob = (nb0 + wn * NBW + i) * 32;
            const int col = min(cob + (lane & 31), p.Cout - 1);
            const float sc = p.scale ? p.scale[col] : 1.f;
            const float sh = p.shift ? p.shift[col] : 0.f;
            if (i) __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r) ot[cd_row(r, lane) * OTP + (lane & 31)] = acc[i][r] * sc + sh;
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int item = lane + q * 64;
                const int row = item >> 2, g = item & 3;
                const int px = pixtab[wm * 32 + row];
                const int co = cob + g * 8;
                if (px >= 0 && co < p.Cout) {
                    const f32x4 lo = *(const f32x4*)(ot + row * OTP + g * 8);
                    const f32x4 hi = *(const f32x4*)(ot + row * OTP + g * 8 + 4);
                    const float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    u16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = elem<T>::bits16(p.relu ? fmaxf(v[e], 0.f) : v[e]);
                    *(u16x8*)(yg + (size_t)px * p.y_cstride + p.y_coff + co) = o;
                }
            }
        }
        return;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NBW; ++i) {
        const int nbg = nb0 + wn * NBW + i;
        const int co = nbg * 32 + (lane & 31);
        if (nbg < p.nblk32 && co < p.Cout) {
            const float sc = p.scale ? p.scale[co] : 1.f;
            const float sh = p.shift ? p.shift[co] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int px = pixtab[wm * 32 + cd_row(r, lane)];
                if (px >= 0) {
                    float v = acc[i][r] * sc + sh;
                    if (p.relu) v = fmaxf(v, 0.f);
                    yg[(size_t)px * p.y_cstride + p.y_coff + co] = elem<T>::from_f32(v);
                }
            }
        }
    }
}

// ---- launch plan: the box.  Work model per workgroup and slab (LDS operations): halo vectors written (x2: a write costs
// about two reads) + pool reads + a fixed part for the two barriers and the exposed latency; fewer than ~512 workgroups
// do not finish sooner than 512.
struct PoolPwPlan { bool ok; int td, th, tw, nbw, wn, groups; long long tiles; };

static PoolPwPlan pool_pw_plan(const step_conv_desc* d, bool s2 = false) {
    PoolPwPlan best; best.ok = false; best.td = best.th = best.tw = 1; best.nbw = 1; best.wn = 1; best.groups = 1; best.tiles = 0;
    const int nblk32 = ceil_div(d->Cout, 32);
    double best_cost = -1;
    for (int wn = 1; wn <= 2; ++wn) {
        const int cap = wn == 1 ? 128 : 64;
        // output blocks per workgroup: up to 4 (128 channels); larger Cout -> channel groups (the pool is recomputed per group)
        const int nbt = nblk32 >= 4 ? 4 : (nblk32 == 3 ? 4 : nblk32);
        if (wn == 2 && (nbt & 1)) continue;
        const int nbw = nbt / wn;
        const int groups = ceil_div(nblk32, nbt);
        for (int td = 1; td <= 4 && td <= d->D; ++td)
            for (int kh_ = 1; kh_ <= d->H; ++kh_) {
                const int th = ceil_div(d->H, kh_);
                if (kh_ > 1 && th == ceil_div(d->H, kh_ - 1)) continue;
                for (int kw_ = 1; kw_ <= d->W; ++kw_) {
                    const int tw = ceil_div(d->W, kw_);
                    if (kw_ > 1 && tw == ceil_div(d->W, kw_ - 1)) continue;
                    if (td * th * tw > cap || (s2 ? td : 1) * th * tw * 4 > PPW_PK * 256) continue;
                    const int halo = s2 ? td * (2 * th + 1) * (2 * tw + 1) : (td + 2) * (th + 2) * (tw + 2);
                    if (halo > PPW_NPIX) continue;
                    const long long tiles = (long long)d->N * ceil_div(d->D, td) * ceil_div(d->H, th) * ceil_div(d->W, tw);
                    const double per = halo * 8.0 + (s2 ? td * th * tw * 4.0 * 9.0 : th * tw * 4.0 * (td + 2) * 9.0) + 4000.0;
                    const long long wgs = tiles * groups;
                    const double cost = (double)(wgs < 512 ? 512 : wgs) * per;
                    if (best_cost < 0 || cost < best_cost) {
                        best_cost = cost;
                        best.ok = true; best.td = td; best.th = th; best.tw = tw; best.nbw = nbw; best.wn = wn; best.groups = groups; best.tiles = tiles;
                    }
                }
            }
    }
    return best;
}

template <typename T, bool S2>
static int pool_pw_launch(const PoolPwPlan& pl, const ConvParams& p, dim3 grid, step_stream_t stream) {
    // halo slabs in flight: 3, or 2 when that wastes less of a short K loop (the loop runs whole rounds of DEPTH slabs)
    const int nslab = ceil_div(p.Cin, 64 / (int)sizeof(T));
    const bool two = ceil_div(nslab, 2) * 2 < ceil_div(nslab, 3) * 3 || pl.nbw == 4;      // (four accumulator tiles + three sets spill)
#define STEP_PPW(NBW_, WN_) do { if (two) STEP_LAUNCH((pool_pw_kernel<T, NBW_, WN_, S2, 2>), grid, dim3(256), stream, p); \
                                 else STEP_LAUNCH((pool_pw_kernel<T, NBW_, WN_, S2, 3>), grid, dim3(256), stream, p); } while (0)
    if (pl.wn == 2) {
        if (pl.nbw == 1) STEP_PPW(1, 2); else STEP_PPW(2, 2);
    } else {
        switch (pl.nbw) {
            case 1: STEP_PPW(1, 1); break;
            case 2: STEP_PPW(2, 1); break;
            default: STEP_PPW(4, 1); break;
        }
    }
#undef STEP_PPW
    return STEP_LAUNCH_CHECK();
}

}  // namespace step

using namespace step;

extern "C" {

static int pool_conv_check(const step_conv_desc* d) {
    if (!d) return STEP_E_NULL;
    if (d->N < 0 || d->D <= 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->Cout <= 0) return STEP_E_SHAPE;
    if (!(d->kd == 1 && d->kh == 1 && d->kw == 1) || d->split) return STEP_E_UNSUPPORTED;
    if (d->x_coff < 0 || d->x_coff + d->Cin > d->x_cstride || d->y_coff < 0 || d->y_coff + d->Cout > d->y_cstride) return STEP_E_SHAPE;
    if (d->dtype != STEP_F32 && d->dtype != STEP_BF16 && d->dtype != STEP_F16) return STEP_E_DTYPE;
    // the kernel keeps 32-bit element offsets of its halo pixels
    if (((unsigned long long)d->N * d->D * d->H * d->W + 1) * (unsigned long long)d->x_cstride >= 0xffffffffULL) return STEP_E_UNSUPPORTED;
    return STEP_OK;
}

// d describes the 1x1x1 conv on the POOLED tensor; s2: the (1,3,3)/(1,2,2) pool in front of it reads x [N, D, Hi, Wi]
static int pool_conv_forward_impl(const step_conv_desc* d, bool s2, int Hi, int Wi, const void* x, const void* w_packed, const float* scale,
                                  const float* shift, void* y, step_stream_t stream) {
    const int chk = pool_conv_check(d);
    if (chk != STEP_OK) return chk;
    if (s2 && (Hi <= 0 || Wi <= 0 || (Hi + 1) / 2 != d->H || (Wi + 1) / 2 != d->W)) return STEP_E_SHAPE;   // ceil(L / 2): step_pool_out_size(L, 3, 2)
    if (s2 && ((unsigned long long)d->N * d->D * Hi * Wi + 1) * (unsigned long long)d->x_cstride >= 0xffffffffULL) return STEP_E_UNSUPPORTED;
    if (d->N == 0) return STEP_OK;
    if (!x || !w_packed || !y) return STEP_E_NULL;
    const int vec = d->dtype == STEP_F32 ? 4 : 8;
    if (d->Cin % vec || d->x_cstride % vec || d->x_coff % vec) return STEP_E_ALIGN;
    if (((uintptr_t)x % 16) || ((uintptr_t)w_packed % 16)) return STEP_E_ALIGN;
    const PoolPwPlan pl = pool_pw_plan(d, s2);
    if (!pl.ok) return STEP_E_UNSUPPORTED;
    ConvParams p;
    p.x = x; p.w = w_packed; p.scale = scale; p.shift = shift; p.res = nullptr; p.y = y; p.y2 = nullptr;
    p.split = 0; p.y2_cstride = 0; p.y2_coff = 0;
    p.N = d->N; p.D = d->D; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Cout = d->Cout;
    p.Hi = s2 ? Hi : d->H; p.Wi = s2 ? Wi : d->W;
    p.x_cstride = d->x_cstride; p.x_coff = d->x_coff; p.y_cstride = d->y_cstride; p.y_coff = d->y_coff;
    p.r_cstride = 0; p.r_coff = 0;
    p.relu = d->relu;
    p.gtd = pl.td; p.gth = pl.th; p.gtw = pl.tw; p.gmode = 0; p.tile0 = 0;
    p.tiles_d = ceil_div(d->D, pl.td); p.tiles_h = ceil_div(d->H, pl.th); p.tiles_w = ceil_div(d->W, pl.tw);
    p.nchunks = ceil_div(d->Cin, CK); p.nchunks32 = p.nchunks;
    p.vec_epi = (d->y_cstride % 8 == 0) && (d->y_coff % 8 == 0) && (d->Cout % 8 == 0) && (((uintptr_t)y) % 16 == 0);
    p.nblk32 = ceil_div(d->Cout, 32);
    p.Mtot = (long long)d->N * d->D * d->H * d->W;
    p.gx = (int)pl.tiles; p.gy = pl.groups;
    const long long tot = pl.tiles * pl.groups;
    const dim3 grid((unsigned)((tot + 7) / 8 * 8));
    if (s2) {
        switch (d->dtype) {
            case STEP_F32: return pool_pw_launch<float, true>(pl, p, grid, stream);
            case STEP_BF16: return pool_pw_launch<bf16_t, true>(pl, p, grid, stream);
            default: return pool_pw_launch<f16_t, true>(pl, p, grid, stream);
        }
    }
    switch (d->dtype) {
        case STEP_F32: return pool_pw_launch<float, false>(pl, p, grid, stream);
        case STEP_BF16: return pool_pw_launch<bf16_t, false>(pl, p, grid, stream);
        default: return pool_pw_launch<f16_t, false>(pl, p, grid, stream);
    }
}

int step_pool3_conv1_forward(const step_conv_desc* d, const void* x, const void* w_packed, const float* scale, const float* shift,
                             void* y, step_stream_t stream) {
    return pool_conv_forward_impl(d, false, 0, 0, x, w_packed, scale, shift, y, stream);
}

int step_pool133s2_conv1_forward(const step_conv_desc* d, int Hi, int Wi, const void* x, const void* w_packed, const float* scale,
                                 const float* shift, void* y, step_stream_t stream) {
    return pool_conv_forward_impl(d, true, Hi, Wi, x, w_packed, scale, shift, y, stream);
}

int step_pool3_conv1_kernel_name(const step_conv_desc* d, char* buf, int buflen) {
    if (!buf || buflen <= 0) return STEP_E_NULL;
    const int chk = pool_conv_check(d);
    if (chk != STEP_OK) return chk;
    const PoolPwPlan pl = pool_pw_plan(d);
    if (!pl.ok) return STEP_E_UNSUPPORTED;
    const char* t = d->dtype == STEP_F32 ? "float" : (d->dtype == STEP_BF16 ? "step::bf16_t" : "step::f16_t");
    snprintf(buf, (size_t)buflen, "void step::pool_pw_kernel<%s, %d, %d>(step::ConvParams)", t, pl.nbw, pl.wn);
    return STEP_OK;
}

}  // extern "C"
