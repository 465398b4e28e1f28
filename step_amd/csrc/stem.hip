// step_amd/csrc/stem.hip -- the I3D stem (7x7x7, stride 2, Cin = 3): stem_igemm_kernel (fp32 parity path),
// stem_tap_kernel, stem_stream_kernel (dense K axis, default for 16-bit storage), their weight packers and the C entry
// points step_stem_*.
#include "conv_common.h"

namespace step {

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
    int tile0;                 // stem_stream_kernel: first pixel tile of this launch (the layer may be launched in two parts)
    int relu;                  // 1: ReLU after the affine (Unit3Dpy); 0: the raw affine output (batch-statistics BatchNorm follows, bn.hip)
    // stem_stream_kernel<.., POOL = true> (step_stem_pool_forward): maxPool3d_2a -- (1,3,3) / (1,2,2), TF padding (0,1) -- taken on the
    // tile while it is still on the chip; y is then the POOLED tensor [N, To, Hp, Wp, C] and the un-pooled stem output never exists
    int Hp, Wp;
    void* rowbuf;              // [N*To][tiles_h][Wo][C]: the first row of every tile (what the tile above still needs), raw stem values
    void* colbuf;              // [N*To][tiles_w][Ho][C]: the first column of every tile
    // stem_stream_kernel<.., U8 = true> (step_stem_pool_forward_u8): x is the decoder's uint8 frames [N,T,H,W,3]; the normalisation of
    // step_clip_from_u8 -- ConvertFromInts(scale) / SubtractMeans / DivideStds, data/augmentations.py:68-111 -- and the rounding to the
    // storage type happen in the frame staging through a 3 x 256-entry table built in the prologue with the same fp32 operations
    int u8_scale; float u8_mean[3], u8_std[3];
#ifdef STEP_PROBE
    unsigned long long* probe;   // tools/timeline_probe.py build only (conv_common.h: probe_mark)
#endif
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
                    float v = acc[i][r] * sc + sh;
                    if (p.relu) v = fmaxf(v, 0.f);
                    const size_t opix = (((size_t)n * p.To + od) * p.Ho + oh) * p.Wo + ow;
                    yg[opix * p.y_cstride + p.y_coff + co] = elem<T>::from_f32(v);
                }
            }
        }
    }
}

constexpr int STP_ROWS = 37, STP_COLS = 40;       // LDS frame of the stream stem: rows x columns of input pixels under a 16x16 output tile

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

// VEC: W % 4 == 0 (whole 4-column quads, 8-byte loads).  A compile-time switch on purpose: with both staging paths inside one loop
// the compiler's wait counts must hold for the element-wise path too (nothing in flight behind it), so every wait for a weight
// tile became vmcnt(0) and drained the frame loads issued behind it.
// POOL: see StemParams::Hp.  A pooled pixel (ph, pw) is the max over stem rows 2ph .. 2ph+2 and columns 2pw .. 2pw+2; a 16x16 tile
// holds everything for 7 of its 8 pooled rows / columns and two of the three rows / columns of the eighth.  The tile writes the max
// over what it HAS to y and its own first row and first column (raw values) to rowbuf / colbuf; stem_pool_fix_kernel then completes
// the pooled pixels on tile seams from those.  Values are post-ReLU (>= +0), so the zero padding of the reference's ConstantPad3d is
// neutral, out-of-image pixels of partial tiles enter as 0, and 16-bit patterns order like signed integers (one v_pk_max_i16 per pair).
constexpr int STS_TPITCH = 144;                     // bytes per pixel of the epilogue's LDS tile: 64 channels x 2 B + 16 B (16 lanes x 16 B at one channel offset spread over all banks)
typedef short s16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ u32x4 pk_max_nonneg16(const u32x4& a, const u32x4& b) {
    return __builtin_bit_cast(u32x4, __builtin_elementwise_max(__builtin_bit_cast(s16x8_t, a), __builtin_bit_cast(s16x8_t, b)));
}

template <typename T, int NB_ = 2, bool VEC = true, bool POOL = false, bool U8 = false>
__global__ __launch_bounds__(256) STEP_WAVES_PER_SIMD(3) void stem_stream_kernel(StemParams p) {
    static_assert(sizeof(T) == 2, "16-bit storage types only");
    static_assert(!U8 || VEC, "uint8 frames: whole 4-column quads only");
    constexpr int FRAME = STS_FRAME, PITCH = STS_PITCH;
    constexpr int NB = NB_, FRAGB = 1024;          // 32-channel blocks per workgroup: 2, or 1 for the tiles of the partial last round
    constexpr int NBREG = 4 * FRAGB;                // one n-block's share of a weight buffer (up to 4 K-steps)
    constexpr int BBUF = NB * NBREG;                // 8 KiB
    constexpr int ITEMS = STP_ROWS * (STP_COLS / 4);   // 4-pixel items per frame
    // Staging is split by wave: waves 0-1 move the weight tiles, waves 2-3 the frames.  A wave's memory counter retires in order, so
    // with every wave doing both a wait for a weight tile (one per four K-steps) also waited for the frame requested just before it
    // -- the frame loads, which go to HBM, never had more than 8 K-steps to land whatever the source distance (measured: a second
    // register set, two frames ahead, changed nothing).  Separate waves have separate counters: the frame waves wait once per frame.
    constexpr int SL = 128;                         // lanes per staging role
    constexpr int FQ = (ITEMS + SL - 1) / SL;       // 3 items per frame lane
    constexpr int NTILES = 21;                      // weight tiles: 3 per frame
    typedef u16x8 frag_t;

    static_assert(!POOL || (NB == 2 && 256 * STS_TPITCH <= 3 * FRAME + 3 * BBUF), "the pooled epilogue's tile lives in the frame / weight rings");
    __shared__ __attribute__((aligned(16))) unsigned char lds[3 * FRAME + 3 * BBUF + NB * 32 * 2 * 4 + (U8 ? 3 * 256 * 2 : 0)];
    unsigned char* const ldsA = lds;
    unsigned char* const ldsB = lds + 3 * FRAME;
    float* const ldsS = (float*)(lds + 3 * FRAME + 3 * BBUF);     // fp32 scale | shift of the workgroup's channels (read in the epilogue)
    unsigned short* const ldsL = (unsigned short*)(lds + 3 * FRAME + 3 * BBUF + NB * 32 * 2 * 4);    // U8: [channel][byte value] -> storage bits

    const int tid = threadIdx.x;
    const int lane = tid & 63;
#ifdef STEP_EMUL
    const int wave = tid >> 6;
#else
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
    const int khalf = lane >> 5;
    const bool isW = wave < 2;                      // wave-uniform staging role
    const int stid = tid & (SL - 1);

    // XCD-aware launch order (see grid_coords in conv_common.h): tiles that are neighbours in (w, h, frame) order --
    // they share input halos and frames -- are consecutive on ONE XCD instead of round-robin over the eight L2s
    STEP_PROBE_IDS(p);
    STEP_PROBE_MARK(p, 0);
    int t = blockIdx.x;
    if ((gridDim.x & 7) == 0) t = (t & 7) * (gridDim.x >> 3) + (t >> 3);
    t += p.tile0;
    const int tw_i = t % p.tiles_w; t /= p.tiles_w;
    const int th_i = t % p.tiles_h; t /= p.tiles_h;
    const int od = t % p.To;
    const int n = t / p.To;
    const int oh0 = th_i * 16, ow0 = tw_i * 16;
    const int nb0 = blockIdx.y * NB;

    const T* xg = (const T*)p.x;
    const unsigned char* wg = (const unsigned char*)p.w;
    constexpr bool vec_ok = VEC;

    // ---- frame staging: LDS col cl <-> input col 2*ow0 - 4 + cl, row r <-> input row 2*oh0 - 2 + r
    struct ItemP { u16x4 c[3]; };                  // planar 16-bit clip: 4 columns of each channel plane
    struct ItemB { unsigned w[3]; };               // uint8 frames: the 12 interleaved bytes of 4 pixels
    typedef typename std::conditional<U8, ItemB, ItemP>::type Item;
    // per-thread item table (frame-invariant): element offset of the item's 4 columns inside a channel plane, or -1
    // when the quad lies outside the image.  With W % 4 == 0 a quad is entirely inside or entirely outside (the tile
    // origin 2*ow0 - 4 is a multiple of 4), so the frame loop needs no division, no branch and no partial quad.
    int qoff[FQ];
#pragma unroll
    for (int q = 0; q < FQ; ++q) {
        const int item = stid + q * SL;
        const int cq = item % (STP_COLS / 4), r = item / (STP_COLS / 4);
        const int ih = 2 * oh0 - 2 + r, iw0 = 2 * ow0 - 4 + cq * 4;
        qoff[q] = (item < ITEMS && ih >= 0 && ih < p.H && iw0 >= 0 && iw0 + 3 < p.W) ? ih * p.W + iw0 : -1;
    }
    const size_t plane_elems = (size_t)p.H * p.W;
    auto load_frame = [&](int f, Item (&it)[FQ]) {
        const int ifr = 2 * od - 2 + f;
        const bool frok = ifr >= 0 && ifr < p.T;                 // workgroup-uniform
        const unsigned short* fbase = (const unsigned short*)xg + ((size_t)n * p.T + (frok ? ifr : 0)) * 3 * plane_elems;
        if constexpr (U8) {
            // [H,W,3] bytes: the quad's 12 bytes are contiguous and 4-byte aligned (W % 4 == 0, quad origins are multiples of 4 pixels)
            const unsigned char* fb8 = (const unsigned char*)p.x + ((size_t)n * p.T + (frok ? ifr : 0)) * 3 * plane_elems;
#pragma unroll
            for (int q = 0; q < FQ; ++q) {
                const unsigned* src = (const unsigned*)(fb8 + (size_t)(qoff[q] >= 0 ? qoff[q] : 0) * 3);
                it[q].w[0] = src[0]; it[q].w[1] = src[1]; it[q].w[2] = src[2];
            }
        } else if constexpr (vec_ok) {
            // raw loads from a clamped (always valid) address; the out-of-image quads are zeroed in store_frame, where the data
            // is needed anyway -- a select on the loaded value HERE made the compiler wait for the loads (vmcnt(0)) before the
            // frame's first MFMA: 2.6 us of exposed latency per frame (tools/timeline_probe.py --stem, round 3)
#pragma unroll
            for (int q = 0; q < FQ; ++q) {
                const unsigned short* src = fbase + (qoff[q] >= 0 ? qoff[q] : 0);
#pragma unroll
                for (int c = 0; c < 3; ++c) it[q].c[c] = *(const u16x4*)(src + c * plane_elems);
            }
        } else {
#pragma unroll
        for (int q = 0; q < FQ; ++q) {                           // W % 4 != 0: element-wise with bounds checks
            const int item = stid + q * SL;
            const int cq = item % (STP_COLS / 4), r = item / (STP_COLS / 4);
            const int ih = 2 * oh0 - 2 + r, iw0 = 2 * ow0 - 4 + cq * 4;
            const bool rowok = item < ITEMS && frok && ih >= 0 && ih < p.H;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                u16x4 v = {0, 0, 0, 0};
                if (rowok) {
                    const unsigned short* src = fbase + c * plane_elems + (size_t)ih * p.W;
#pragma unroll
                    for (int a = 0; a < 4; ++a)
                        if (iw0 + a >= 0 && iw0 + a < p.W) v[a] = src[iw0 + a];
                }
                it[q].c[c] = v;
            }
        }
        }
    };
    auto store_frame = [&](int slotoff, int f, const Item (&it)[FQ]) {
        const int ifr = 2 * od - 2 + f;
        const bool frok = ifr >= 0 && ifr < p.T;                 // workgroup-uniform
#pragma unroll
        for (int q = 0; q < FQ; ++q) {
            const int item = stid + q * SL;
            if (item < ITEMS) {
                const int cq = item % (STP_COLS / 4), r = item / (STP_COLS / 4);
                unsigned char* dst = ldsA + slotoff + r * PITCH + cq * 24;
                const unsigned short keep = (!vec_ok || (frok && qoff[q] >= 0)) ? 0xffffu : 0u;    // (the element-wise path zeroed on load)
                u16x4 v0, v1, v2;
                if constexpr (U8) {
                    // the LDS element stream [column][3 channels] IS the byte order of the frame: element e <- table[e % 3][byte e]
                    unsigned short val[12];
#pragma unroll
                    for (int e = 0; e < 12; ++e) val[e] = ldsL[(e % 3) * 256 + ((it[q].w[e >> 2] >> (8 * (e & 3))) & 0xffu)];
                    v0 = u16x4{val[0], val[1], val[2], val[3]}; v1 = u16x4{val[4], val[5], val[6], val[7]}; v2 = u16x4{val[8], val[9], val[10], val[11]};
                } else {
                    v0 = u16x4{it[q].c[0][0], it[q].c[1][0], it[q].c[2][0], it[q].c[0][1]};
                    v1 = u16x4{it[q].c[1][1], it[q].c[2][1], it[q].c[0][2], it[q].c[1][2]};
                    v2 = u16x4{it[q].c[2][2], it[q].c[0][3], it[q].c[1][3], it[q].c[2][3]};
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) { v0[e] &= keep; v1[e] &= keep; v2[e] &= keep; }
                *(u16x4*)dst = v0;
                *(u16x4*)(dst + 8) = v1;
                *(u16x4*)(dst + 16) = v2;
            }
        }
    };

    // ---- weights: a weight lane moves two 16-byte vectors per n-block of a tile (K-steps kstep0 .. kstep0+3;
    //      the 3-K-step tiles carry one K-step of the next tile along, never read)
    const unsigned char* wthr[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i)
        wthr[i] = wg + ((size_t)min(nb0 + i, p.nblk32 - 1) * STS_KSTEPS) * FRAGB + stid * 16;
    struct BReg { u32x4 v[NB][2]; };
    auto load_B = [&](int tile) {
        tile = min(tile, NTILES - 1);
        const int ks0 = (tile / 3) * 11 + (tile % 3) * 4;
        BReg r;
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) r.v[i][j] = *(const u32x4*)(wthr[i] + (size_t)ks0 * FRAGB + j * (SL * 16));
        return r;
    };
    auto store_B = [&](int buf, const BReg& r) {
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) *(u32x4*)(ldsB + buf * BBUF + i * NBREG + j * (SL * 16) + stid * 16) = r.v[i][j];
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
        for (int i = 0; i < NB; ++i) {          // weights as the FIRST operand: transposed accumulators (a lane owns one pixel), see the epilogue
            mma_k16(fb[SET][i], fa[SET][0], acc[0][i], T());
            mma_k16(fb[SET][i], fa[SET][1], acc[1][i], T());
        }
    };

    // ---- prologue: frames 0 and 1, weight tiles 0 and 1, tile 2 in flight
    // (every request first, then the LDS stores: one memory round trip instead of four -- frame 0, frame 1, weight tiles 0 and 1
    // used to be loaded and parked one after the other)
    // The whole K loop exists twice, once per staging role (compile-time W): inside ONE loop the role tests merge control flow at
    // every staging point, and the values requested in a branch then reach the loop-carried registers through copies that wait for
    // the loads -- measured in the ISA, not guessed.  Both bodies execute the same barriers.
    if constexpr (U8) {
        // the normalisation table, 3 entries per thread: the fp32 operations of clip_from_u8_kernel in the same order (no FMA contraction),
        // then the storage rounding -- exactly the element the two-launch path would have read from the converted clip
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float v = (float)tid;
            if (p.u8_scale == 1) v = __fdiv_rn(v, 255.f);
            else if (p.u8_scale == 2) v = __fsub_rn(__fdiv_rn(__fmul_rn(v, 2.f), 255.f), 1.f);
            v = __fdiv_rn(__fsub_rn(v, p.u8_mean[k]), p.u8_std[k]);
            ldsL[k * 256 + tid] = elem<T>::bits16(v);
        }
        __syncthreads();
    }
    auto run_role = [&](auto rolec) {
    constexpr bool W = decltype(rolec)::value;
    Item fr[FQ], fr1[FQ];
    Item frS[FQ];                                  // the frame in flight during the K loop
    BReg R, R1;
    if constexpr (W) {
        float ss_sc = 1.f, ss_sh = 0.f;
        R = load_B(0);
        R1 = load_B(1);
        if (tid < NB * 32) {
            const int co = min(nb0 * 32 + tid, p.Cout - 1);
            if (p.scale) ss_sc = p.scale[co];
            if (p.shift) ss_sh = p.shift[co];
        }
        store_B(0, R);
        store_B(1, R1);
        if (tid < NB * 32) { ldsS[tid] = ss_sc; ldsS[NB * 32 + tid] = ss_sh; }
        R = load_B(2);
    } else {
        load_frame(0, fr);
        load_frame(1, fr1);
        load_frame(2, frS);                        // stays in registers until frame 0's K-steps park it
        store_frame(0, 0, fr);
        store_frame(FRAME, 1, fr1);
    }
    __syncthreads();
    STEP_PROBE_MARK(p, 1);
    read_frags(std::integral_constant<int, 0>(), std::integral_constant<int, 0>(), 0);

    int cur = 0, nxt = FRAME, nn = 2 * FRAME;          // slot byte offsets of frames kd, kd+1, kd+2
    // one frame = 11 K-steps; P = parity of the frame's first K-step (11 is odd, so it alternates)
    // The frame waves park frame kd + 2 (requested one frame = 11 K-steps earlier) before the second barrier of frame kd and request
    // frame kd + 3 into the same registers right behind it.
    // LD / ST (compile time): frame kd + 3 exists and is requested / frame kd + 2 exists and is parked during this frame.  Not
    // run-time tests: a conditional load in the loop forces the waits behind it to a full drain (the compiler's counts must hold
    // for the path without the load).
    auto frame_iter = [&](auto pc, auto ldc, auto stc, int kd) {
        constexpr int P = decltype(pc)::value;
        constexpr bool LD = decltype(ldc)::value, ST = decltype(stc)::value;
#define STS_KSTEP(J)                                                                                                  \
        {                                                                                                             \
            if (J < 10) read_frags(std::integral_constant<int, (P + J + 1) & 1>(), std::integral_constant<int, (J + 1) % 11>(), cur); \
            else        read_frags(std::integral_constant<int, (P + J + 1) & 1>(), std::integral_constant<int, 0>(), nxt);            \
            mma_all(std::integral_constant<int, (P + J) & 1>());                                                      \
        }
#define STS_TILE_END(TI)                                                                                              \
        {                                                                                                             \
            if constexpr (W) {                                                                                        \
                store_B((TI + 2) % 3, R);              /* tile 3*kd + TI + 2 -> its home buffer */                  \
                R = load_B(3 * kd + TI + 3);                                                                          \
            }                                                                                                         \
        }
        STS_KSTEP(0) STS_KSTEP(1) STS_KSTEP(2) STS_KSTEP(3)
        STS_TILE_END(0)
        __syncthreads();
        STS_KSTEP(4) STS_KSTEP(5) STS_KSTEP(6) STS_KSTEP(7)
        STS_TILE_END(1)
        if constexpr (ST && !W) {
            STEP_SCHED_BARRIER();                      // (the scheduler pulled the repacking of the frame up to the frame's first K-steps -- and the wait for its loads with it)
            store_frame(nn, kd + 2, frS);              // the slot frame kd-1 left (last read before the previous frame's last barrier)
            if constexpr (LD) load_frame(kd + 3, frS);
        }
        __syncthreads();
        STS_KSTEP(8) STS_KSTEP(9) STS_KSTEP(10)
        STS_TILE_END(2)
        __syncthreads();
#undef STS_KSTEP
#undef STS_TILE_END
        const int tmp = cur; cur = nxt; nxt = nn; nn = tmp;
    };
#pragma unroll 1
    for (int kd = 0; kd < 4; kd += 2) {
        frame_iter(std::integral_constant<int, 0>(), std::true_type(), std::true_type(), kd);
        frame_iter(std::integral_constant<int, 1>(), std::true_type(), std::true_type(), kd + 1);
        STEP_PROBE_MARK(p, 5 + kd / 2);                 // slots 5, 6: frames 0-1, 2-3 done
    }
    frame_iter(std::integral_constant<int, 0>(), std::false_type(), std::true_type(), 4);
    frame_iter(std::integral_constant<int, 1>(), std::false_type(), std::false_type(), 5);
    STEP_PROBE_MARK(p, 7);                              // frames 4-5 done
    frame_iter(std::integral_constant<int, 0>(), std::false_type(), std::false_type(), 6);
    };
    if (isW) run_role(std::true_type());
    else run_role(std::false_type());
    STEP_PROBE_MARK(p, 2);

    // ---- epilogue: affine + ReLU and 16-byte stores straight from registers.  The accumulators are transposed (lane l owns pixel
    // (l & 31) of each of its two row blocks and, in registers 4g .. 4g+3, the channels 8g + 4 (l >> 5) + {0..3}); one
    // v_permlane32_swap per packed dword pair hands every lane 8 consecutive channels of its pixel (conv_tap_kernel.h: the LDS
    // transpose this replaces -- two passes of 32 ds_write_b32 + barrier + read-out -- cost a third of the epilogue there).
    T* yg = (T*)p.y;
    const bool vec_epi = (p.y_cstride % 8 == 0) && (p.y_coff % 8 == 0) && (p.Cout % 8 == 0) && (((uintptr_t)p.y) % 16 == 0);
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        const int oh = oh0 + wave * 4 + mb + 2 * ((lane & 31) >> 4), ow = ow0 + (lane & 15);
        const bool okp = oh < p.Ho && ow < p.Wo;
        const size_t opix = okp ? (((size_t)n * p.To + od) * p.Ho + oh) * p.Wo + ow : 0;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            unsigned d[4][2];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 sc = *(const f32x4*)(ldsS + i * 32 + 8 * g + 4 * khalf);
                const f32x4 sh = *(const f32x4*)(ldsS + NB * 32 + i * 32 + 8 * g + 4 * khalf);
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = acc[mb][i][4 * g + e] * sc[e] + sh[e];
                    if (p.relu) v[e] = fmaxf(v[e], 0.f);
                }
                if (!vec_epi && !POOL) {                      // channel counts / offsets off the 16-byte grid: element stores
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int co = (nb0 + i) * 32 + 8 * g + 4 * khalf + e;
                        if (okp && co < p.Cout) yg[opix * p.y_cstride + p.y_coff + co] = elem<T>::from_f32(v[e]);
                    }
                }
                d[g][0] = (unsigned)elem<T>::bits16(v[0]) | ((unsigned)elem<T>::bits16(v[1]) << 16);
                d[g][1] = (unsigned)elem<T>::bits16(v[2]) | ((unsigned)elem<T>::bits16(v[3]) << 16);
            }
            if constexpr (POOL) {
                // the tile goes to LDS (the rings are free: the K loop ended in a barrier), pixel-major at a 144-byte pitch
                const int tpix = (wave * 4 + mb + 2 * ((lane & 31) >> 4)) * 16 + (lane & 15);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    lane32_swap(d[2 * h][0], d[2 * h + 1][0]);
                    lane32_swap(d[2 * h][1], d[2 * h + 1][1]);
                    u32x4 o = {d[2 * h][0], d[2 * h][1], d[2 * h + 1][0], d[2 * h + 1][1]};
                    if (!okp) o = u32x4{0u, 0u, 0u, 0u};
                    *(u32x4*)(lds + tpix * STS_TPITCH + (i * 32 + 16 * h + 8 * khalf) * 2) = o;
                }
            } else if (vec_epi) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    lane32_swap(d[2 * h][0], d[2 * h + 1][0]);
                    lane32_swap(d[2 * h][1], d[2 * h + 1][1]);
                    const int co = (nb0 + i) * 32 + 16 * h + 8 * khalf;
                    if (okp && co < p.Cout) {
                        const u32x4 o = {d[2 * h][0], d[2 * h][1], d[2 * h + 1][0], d[2 * h + 1][1]};
                        *(u32x4*)(yg + opix * p.y_cstride + p.y_coff + co) = o;
                    }
                }
            }
        }
    }
    if constexpr (POOL) {
        __syncthreads();
        const size_t plane = (size_t)n * p.To + od;
        // 8 x 8 pooled pixels x 8 channel vectors = 512 items, two per thread; vector fastest (8 lanes = one pixel's 128 bytes)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int item = tid + 256 * k;
            const int v = item & 7, q = (item >> 3) & 7, j = item >> 6;
            const unsigned char* base = lds + ((2 * j) * 16 + 2 * q) * STS_TPITCH + v * 16;
            u32x4 m = *(const u32x4*)base;
#pragma unroll
            for (int dr = 0; dr < 3; ++dr)
#pragma unroll
                for (int dc = 0; dc < 3; ++dc) {
                    if (dr == 0 && dc == 0) continue;
                    if (2 * j + dr < 16 && 2 * q + dc < 16) m = pk_max_nonneg16(m, *(const u32x4*)(base + (dr * 16 + dc) * STS_TPITCH));
                }
            const int ph = (oh0 >> 1) + j, pw = (ow0 >> 1) + q;
            if (ph < p.Hp && pw < p.Wp)
                *(u32x4*)(yg + ((plane * p.Hp + ph) * p.Wp + pw) * p.y_cstride + p.y_coff + v * 8) = m;
        }
        // the tile's first row -> rowbuf, first column -> colbuf (raw values; a tile in the first tile row / column has no reader)
        {
            const int v = tid & 7, e = (tid >> 3) & 15;
            unsigned short* rb = (unsigned short*)p.rowbuf;
            unsigned short* cb = (unsigned short*)p.colbuf;
            if (tid < 128) {
                if (th_i > 0 && ow0 + e < p.Wo)
                    *(u32x4*)(rb + ((plane * p.tiles_h + th_i) * p.Wo + ow0 + e) * (size_t)p.Cout + v * 8) = *(const u32x4*)(lds + e * STS_TPITCH + v * 16);
            } else {
                if (tw_i > 0 && oh0 + e < p.Ho)
                    *(u32x4*)(cb + ((plane * p.tiles_w + tw_i) * p.Ho + oh0 + e) * (size_t)p.Cout + v * 8) = *(const u32x4*)(lds + (e * 16) * STS_TPITCH + v * 16);
            }
        }
    }
#ifdef STEP_PROBE
    STEP_PROBE_MARK(p, 3);
    __builtin_amdgcn_s_waitcnt(0);                    // every store acknowledged
    probe_clock_end(p.probe);
    STEP_PROBE_MARK(p, 4);
#endif
}

// Completes the pooled pixels on tile seams of stem_stream_kernel<.., POOL> (see there): one thread per (plane, seam pixel, 8-channel
// vector).  Row seams: pooled row 8t - 1 lacks stem row 16t = the first row of tile row t (rowbuf); the pixels that are ALSO on a column
// seam take the two colbuf rows they lack here too, so that exactly one thread updates any pooled pixel.  Column seams: pooled column
// 8s - 1 lacks stem column 16s (colbuf), rows 2ph .. 2ph+2 -- all inside one tile row unless ph is a seam row (handled above).
__global__ __launch_bounds__(256) void stem_pool_fix_kernel(StemParams p) {
    const int nbr = p.tiles_h - 1, nbc = p.tiles_w - 1;
    const long long per_plane = ((long long)nbr * p.Wp + (long long)nbc * p.Hp) * (p.Cout / 8);
    const long long total = per_plane * p.N * p.To;
    const int V = p.Cout / 8;
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
        if (it < (long long)nbr * p.Wp) { row_item = true; ph = 8 * ((int)(it / p.Wp) + 1) - 1; pw = (int)(it % p.Wp); }
        else { it -= (long long)nbr * p.Wp; row_item = false; pw = 8 * ((int)(it / p.Hp) + 1) - 1; ph = (int)(it % p.Hp); }
        if (ph >= p.Hp || pw >= p.Wp) continue;
        const bool ph_seam = ((ph + 1) & 7) == 0 && ((ph + 1) >> 3) <= nbr;
        const bool pw_seam = ((pw + 1) & 7) == 0 && ((pw + 1) >> 3) <= nbc;
        if (!row_item && ph_seam) continue;                       // a corner: the row pass owns it
        unsigned short* yp = yg + ((plane * p.Hp + ph) * p.Wp + pw) * (size_t)p.y_cstride + p.y_coff + v * 8;
        u32x4 m = *(const u32x4*)yp;
        if (row_item) {
            const int t = (ph + 1) >> 3;
            for (int dc = 0; dc < 3; ++dc) {
                const int c = 2 * pw + dc;
                if (c < p.Wo) m = pk_max_nonneg16(m, *(const u32x4*)(rb + ((plane * p.tiles_h + t) * p.Wo + c) * (size_t)p.Cout + v * 8));
            }
            if (pw_seam) {
                const int s = (pw + 1) >> 3;
                for (int dr = 0; dr < 2; ++dr)
                    m = pk_max_nonneg16(m, *(const u32x4*)(cb + ((plane * p.tiles_w + s) * p.Ho + 2 * ph + dr) * (size_t)p.Cout + v * 8));
            }
        } else {
            const int s = (pw + 1) >> 3;
            for (int dr = 0; dr < 3; ++dr) {
                const int r = 2 * ph + dr;
                if (r < p.Ho) m = pk_max_nonneg16(m, *(const u32x4*)(cb + ((plane * p.tiles_w + s) * p.Ho + r) * (size_t)p.Cout + v * 8));
            }
        }
        *(u32x4*)yp = m;
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
    const long long tiles = (long long)p.N * p.To * p.tiles_h * p.tiles_w;
    const int groups = ceil_div(p.nblk32, 2);
    // (a separate NB = 1 launch for the partial last round, as conv_forward_t does for conv3d_2c, was built and measured SLOWER
    // here -- 263 -> 268 us: with three resident workgroups per CU the rounds are not in step and the tail is already smeared
    // out -- and removed again)
    p.tile0 = 0;
#ifdef STEP_PROBE
    p.probe = g_probe_buf;
#endif
    dim3 grid((unsigned)tiles, (unsigned)groups);
    if (p.W % 4 == 0) STEP_LAUNCH((stem_stream_kernel<T, 2, true>), grid, dim3(256), stream, p);
    else STEP_LAUNCH((stem_stream_kernel<T, 2, false>), grid, dim3(256), stream, p);
    return STEP_LAUNCH_CHECK();
}


}  // namespace step

using namespace step;

extern "C" {

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
                      const float* shift, int relu, int Cout, void* y, int y_cstride, int y_coff, step_stream_t stream) {
    if (N < 0 || T <= 0 || H <= 0 || W <= 0 || Cout <= 0) return STEP_E_SHAPE;
    if (y_coff < 0 || y_coff + Cout > y_cstride) return STEP_E_SHAPE;
    if (N == 0) return STEP_OK;
    if (!x || !w_packed || !y) return STEP_E_NULL;
    if (((uintptr_t)x % 16) || ((uintptr_t)w_packed % 16)) return STEP_E_ALIGN;
    StemParams p;
    p.x = x; p.w = w_packed; p.scale = scale; p.shift = shift; p.y = y;
    p.N = N; p.T = T; p.H = H; p.W = W; p.tile0 = 0; p.relu = relu != 0;
    p.To = (T + 5 - 7) / 2 + 1; p.Ho = (H + 5 - 7) / 2 + 1; p.Wo = (W + 5 - 7) / 2 + 1;
    if (p.To <= 0 || p.Ho <= 0 || p.Wo <= 0) return STEP_E_SHAPE;
    p.Cout = Cout; p.y_cstride = y_cstride; p.y_coff = y_coff;
    p.tiles_h = ceil_div(p.Ho, STEM_TH); p.tiles_w = ceil_div(p.Wo, STEM_TW);
    p.nblk32 = ceil_div(Cout, 32);
    const int ov = conv_impl_override();
    switch (dtype) {
        case STEP_F32: return stem_forward_t<float>(p, stream);
        // STEP_OPT_CONV_IMPL = 0 selects the tiled stem (the fp32 kernel) for the 16-bit types too (tests); default = dense-K stream stem
        case STEP_BF16: return ov == 0 ? stem_forward_t<bf16_t>(p, stream) : stem_stream_forward_t<bf16_t>(p, stream);
        case STEP_F16: return ov == 0 ? stem_forward_t<f16_t>(p, stream) : stem_stream_forward_t<f16_t>(p, stream);
    }
    return STEP_E_DTYPE;
}


static bool stem_pool_supported(int dtype, int N, int T, int H, int W, int Cout) {
    return (dtype == STEP_BF16 || dtype == STEP_F16) && N > 0 && T > 0 && H > 0 && W > 0 && (W % 4) == 0 && Cout == 64 && conv_impl_override() != 0;
}

size_t step_stem_pool_workspace_bytes(int dtype, int N, int T, int H, int W, int Cout) {
    if (!stem_pool_supported(dtype, N, T, H, W, Cout)) return 0;
    const int To = (T - 2) / 2 + 1, Ho = (H - 2) / 2 + 1, Wo = (W - 2) / 2 + 1;
    const size_t planes = (size_t)N * To;
    return planes * ((size_t)ceil_div(Ho, 16) * Wo + (size_t)ceil_div(Wo, 16) * Ho) * Cout * 2 + 32;
}

// parts: 1 = the stem launch (tiles + their first rows / columns), 2 = the seam pass, 3 = both
static int stem_pool_forward_impl(int dtype, const void* x, int N, int T, int H, int W, const void* w_packed, const float* scale, const float* shift,
                                  int Cout, void* y, int y_cstride, int y_coff, void* ws, size_t ws_bytes, int parts, step_stream_t stream) {
    if (N < 0 || T <= 0 || H <= 0 || W <= 0 || Cout <= 0) return STEP_E_SHAPE;
    if (y_coff < 0 || y_coff + Cout > y_cstride) return STEP_E_SHAPE;
    if (N == 0) return STEP_OK;
    if (!stem_pool_supported(dtype, N, T, H, W, Cout)) return STEP_E_UNSUPPORTED;
    if (!x || !w_packed || !y || !ws) return STEP_E_NULL;
    if (((uintptr_t)x % 16) || ((uintptr_t)w_packed % 16) || ((uintptr_t)y % 16) || ((uintptr_t)ws % 16) || (y_cstride % 8) || (y_coff % 8)) return STEP_E_ALIGN;
    if (ws_bytes < step_stem_pool_workspace_bytes(dtype, N, T, H, W, Cout)) return STEP_E_SHAPE;
    StemParams p;
    p.x = x; p.scale = scale; p.shift = shift; p.y = y;
    p.N = N; p.T = T; p.H = H; p.W = W; p.tile0 = 0; p.relu = 1;
    p.To = (T + 5 - 7) / 2 + 1; p.Ho = (H + 5 - 7) / 2 + 1; p.Wo = (W + 5 - 7) / 2 + 1;
    if (p.To <= 0 || p.Ho <= 0 || p.Wo <= 0) return STEP_E_SHAPE;
    p.Cout = Cout; p.y_cstride = y_cstride; p.y_coff = y_coff;
    p.nblk32 = ceil_div(Cout, 32);
    p.tiles_h = ceil_div(p.Ho, 16); p.tiles_w = ceil_div(p.Wo, 16);
    p.Hp = step_pool_out_size(p.Ho, 3, 2); p.Wp = step_pool_out_size(p.Wo, 3, 2);
    p.rowbuf = ws;
    p.colbuf = (unsigned char*)ws + (((size_t)N * p.To * p.tiles_h * p.Wo * Cout * 2 + 15) / 16) * 16;
#ifdef STEP_PROBE
    p.probe = g_probe_buf;
#endif
    const long long tiles = (long long)p.N * p.To * p.tiles_h * p.tiles_w;
    dim3 grid((unsigned)tiles, 1);
    if (parts & 1) {
        if (dtype == STEP_BF16) {
            p.w = (const bf16_t*)w_packed + stem_stream_offset(Cout);
            STEP_LAUNCH((stem_stream_kernel<bf16_t, 2, true, true>), grid, dim3(256), stream, p);
        } else {
            p.w = (const f16_t*)w_packed + stem_stream_offset(Cout);
            STEP_LAUNCH((stem_stream_kernel<f16_t, 2, true, true>), grid, dim3(256), stream, p);
        }
    }
    if ((parts & 2) && (p.tiles_h > 1 || p.tiles_w > 1)) {
        const long long items = ((long long)(p.tiles_h - 1) * p.Wp + (long long)(p.tiles_w - 1) * p.Hp) * (Cout / 8) * p.N * p.To;
        STEP_LAUNCH(stem_pool_fix_kernel, dim3(flat_grid(items, 256)), dim3(256), stream, p);
    }
    return STEP_LAUNCH_CHECK();
}

int step_stem_pool_forward(int dtype, const void* x, int N, int T, int H, int W, const void* w_packed, const float* scale, const float* shift,
                           int Cout, void* y, int y_cstride, int y_coff, void* ws, size_t ws_bytes, step_stream_t stream) {
    return stem_pool_forward_impl(dtype, x, N, T, H, W, w_packed, scale, shift, Cout, y, y_cstride, y_coff, ws, ws_bytes, 3, stream);
}

int step_stem_pool_forward_tiles(int dtype, const void* x, int N, int T, int H, int W, const void* w_packed, const float* scale, const float* shift,
                                 int Cout, void* y, int y_cstride, int y_coff, void* ws, size_t ws_bytes, step_stream_t stream) {
    return stem_pool_forward_impl(dtype, x, N, T, H, W, w_packed, scale, shift, Cout, y, y_cstride, y_coff, ws, ws_bytes, 1, stream);
}

int step_stem_pool_finish(int dtype, const void* x, int N, int T, int H, int W, int Cout, void* y, int y_cstride, int y_coff, void* ws, size_t ws_bytes,
                          step_stream_t stream) {
    // (x is only checked for alignment / presence by the shared argument checks: pass the clip the tiles launch read)
    return stem_pool_forward_impl(dtype, x, N, T, H, W, x, nullptr, nullptr, Cout, y, y_cstride, y_coff, ws, ws_bytes, 2, stream);
}

int step_stem_pool_forward_u8(int dtype, const unsigned char* frames, int N, int T, int H, int W, int u8_scale, const float* mean3, const float* std3,
                              const void* w_packed, const float* scale, const float* shift, int Cout, void* y, int y_cstride, int y_coff,
                              void* ws, size_t ws_bytes, step_stream_t stream) {
    if (N < 0 || T <= 0 || H <= 0 || W <= 0 || Cout <= 0 || u8_scale < 0 || u8_scale > 2) return STEP_E_SHAPE;
    if (y_coff < 0 || y_coff + Cout > y_cstride) return STEP_E_SHAPE;
    if (N == 0) return STEP_OK;
    if (!stem_pool_supported(dtype, N, T, H, W, Cout)) return STEP_E_UNSUPPORTED;
    if (!frames || !w_packed || !y || !ws) return STEP_E_NULL;
    if (((uintptr_t)frames % 4) || ((uintptr_t)w_packed % 16) || ((uintptr_t)y % 16) || ((uintptr_t)ws % 16) || (y_cstride % 8) || (y_coff % 8)) return STEP_E_ALIGN;
    if (ws_bytes < step_stem_pool_workspace_bytes(dtype, N, T, H, W, Cout)) return STEP_E_SHAPE;
    StemParams p;
    p.x = frames; p.scale = scale; p.shift = shift; p.y = y;
    p.N = N; p.T = T; p.H = H; p.W = W; p.tile0 = 0; p.relu = 1;
    p.To = (T + 5 - 7) / 2 + 1; p.Ho = (H + 5 - 7) / 2 + 1; p.Wo = (W + 5 - 7) / 2 + 1;
    if (p.To <= 0 || p.Ho <= 0 || p.Wo <= 0) return STEP_E_SHAPE;
    p.Cout = Cout; p.y_cstride = y_cstride; p.y_coff = y_coff;
    p.nblk32 = ceil_div(Cout, 32);
    p.tiles_h = ceil_div(p.Ho, 16); p.tiles_w = ceil_div(p.Wo, 16);
    p.Hp = step_pool_out_size(p.Ho, 3, 2); p.Wp = step_pool_out_size(p.Wo, 3, 2);
    p.rowbuf = ws;
    p.colbuf = (unsigned char*)ws + (((size_t)N * p.To * p.tiles_h * p.Wo * Cout * 2 + 15) / 16) * 16;
    p.u8_scale = u8_scale;
    for (int c = 0; c < 3; ++c) { p.u8_mean[c] = mean3 ? mean3[c] : 0.f; p.u8_std[c] = std3 ? std3[c] : 1.f; }     // host pointers (3 floats)
#ifdef STEP_PROBE
    p.probe = g_probe_buf;
#endif
    const long long tiles = (long long)p.N * p.To * p.tiles_h * p.tiles_w;
    dim3 grid((unsigned)tiles, 1);
    if (dtype == STEP_BF16) {
        p.w = (const bf16_t*)w_packed + stem_stream_offset(Cout);
        STEP_LAUNCH((stem_stream_kernel<bf16_t, 2, true, true, true>), grid, dim3(256), stream, p);
    } else {
        p.w = (const f16_t*)w_packed + stem_stream_offset(Cout);
        STEP_LAUNCH((stem_stream_kernel<f16_t, 2, true, true, true>), grid, dim3(256), stream, p);
    }
    if (p.tiles_h > 1 || p.tiles_w > 1) {
        const long long items = ((long long)(p.tiles_h - 1) * p.Wp + (long long)(p.tiles_w - 1) * p.Hp) * (Cout / 8) * p.N * p.To;
        STEP_LAUNCH(stem_pool_fix_kernel, dim3(flat_grid(items, 256)), dim3(256), stream, p);
    }
    return STEP_LAUNCH_CHECK();
}

int step_stem_kernel_name(int dtype, char* buf, int buflen) {
    if (!buf || buflen <= 0) return STEP_E_NULL;
    const int ov = conv_impl_override();
    const char* t = dtype == STEP_F32 ? "float" : (dtype == STEP_BF16 ? "step::bf16_t" : "step::f16_t");
    if (dtype != STEP_F32 && dtype != STEP_BF16 && dtype != STEP_F16) return STEP_E_DTYPE;
    if (dtype == STEP_F32 || ov == 0) snprintf(buf, (size_t)buflen, "void step::stem_igemm_kernel<%s, 2>(step::StemParams)", t);
    else snprintf(buf, (size_t)buflen, "void step::stem_stream_kernel<%s, 2, true, false, false>(step::StemParams)", t);
    return STEP_OK;
}


}  // extern "C"
