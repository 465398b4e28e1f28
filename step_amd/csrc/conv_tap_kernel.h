// step_amd/csrc/conv_tap_kernel.h -- the pipelined 8-wave implicit-GEMM conv kernel (3x3x3 and 1x3x3 windows) and its
// launcher template; included by conv_tap_f32.hip / conv_tap_bf16.hip / conv_tap_f16.hip (one storage type per
// translation unit, so the three build in parallel).
#pragma once
#include "conv_common.h"
#include "conv_pw_kernel.h"

namespace step {

// ============================================================================================
// conv_tap_kernel -- the heavy 3x3x3 / 1x3x3 path.
//
// 512 threads = 8 wavefronts own a 256-pixel x (64*NB)-channel output tile.  Waves are arranged
// 4 (pixels) x 2 (channels); each wave accumulates 2 x NB 32x32 MFMA tiles (64 px x 32*NB ch).
//   * A: a 64-byte-per-pixel slab (32 channels of 16-bit data, 16 of fp32) of the input halo tile
//     ([kd][TH+kh-1][TW+kw-1] pixels) is staged into LDS once per slab; all taps read it at shifted
//     bases (im2col-free).  Pixels sit at an 80-byte pitch (64 B + 16 B pad): consecutive pixels
//     rotate through the LDS banks without an XOR swizzle and a tap shift is a plain byte offset.
//   * B: the weights of TPS taps x slab x tile-channels (TPS * 4*NB KiB per pipeline step, already in MFMA fragment
//     order in global memory, so the copy is linear) go global -> two register sets -> three LDS step buffers: a
//     step's weights are loaded four steps before they are used and written to LDS two steps before; fragments are
//     double-buffered in registers per tap, one barrier per step.  Every B fragment read from LDS feeds 2 MFMAs and
//     every A fragment NB MFMAs (a k16 step costs 2 + NB ds_read_b128 for 2*NB MFMAs), and the weights cross the
//     L2 -> CU path once per 256 pixels instead of once per 32.
//   * Launch order, tile shapes and the bank-conflict-free row-to-column maps: see grid_coords (conv_common.h) and
//     the comments at TDL / HWPAD / tile_col below.
// MB = 32-pixel accumulator rows per wave: MB = 2 -> 8 waves (4 x 2, 512 threads, 2 waves per SIMD) is the only
// instantiated form (a 4-wave MB = 4 form -- 30 % less LDS traffic per MFMA -- spills at NB >= 2).

// WV = wavefronts per workgroup.  WV = 8: the 256-pixel tile above, ONE workgroup per CU (up to ~150 KiB of LDS).  WV = 4:
// the same per-wave work (64 pixels x 32*NB channels) on a 128-pixel tile with one tap per barrier, sized so that TWO
// workgroups are resident per CU (<= 80 KiB of LDS each): they are not lock-stepped by each other's barriers, and one
// workgroup's prologue / slab switch / epilogue (about a fifth of a workgroup's lifetime on conv3d_2c, all of it
// exposed with a single resident workgroup) overlaps the other's matrix work.
//
// PH = 1 / 2 (16-bit storage, 8 waves, two taps per step): the TWO-PHASE form.  The matrix pipe is per SIMD and a single wave
// issuing back-to-back keeps it busy (32 cycles per v_mfma_f32_32x32x16), but a wave is in-order: in the classic form both
// waves of a SIMD read their fragments, wait, multiply and meet at the step's barrier at the same moments, so the pipe idles
// whenever they wait (measured: matrix time and the LDS / barrier skeleton ADD instead of overlapping, MFMA busy 37 %).
// Here a step is two phases separated by barriers -- L: all of a step's fragments LDS -> registers (20 ds_read_b128 at
// NB = 3) plus the weight hand-over, C: the step's 8 * NB MFMAs back to back out of registers -- and the two wave groups
// of the workgroup (one wave of each group per SIMD) run them in ANTI-PHASE: group 1 passes one extra barrier before a
// slab's steps and group 0 one after them, so while one wave of a SIMD multiplies, its partner loads.  The groups are the
// two channel halves of the tile (wn), so each group owns the weight ring of its half and stages it with its own 256
// threads; the halo slab is shared and re-staged between slabs with the groups realigned.  No fragment double-buffering
// (a wave's L and C phases alternate) -- same accumulation order as the classic form, bit-identical results.
// PH = 1: group = wave / 4 (waves w and w + 4 share a SIMD), PH = 2: group = wave & 1.
// PRE (two-phase form only): the conv's input is produced on the fly -- a pointwise 64 -> 64 conv + affine + ReLU (conv3d_2b in
// front of conv3d_2c, models/i3dpt.py:207-209) applied to every halo pixel while the halo is staged: the wave reads the raw pixels
// (p.x: the tensor BEFORE the pointwise layer) straight from global memory in MFMA operand layout, multiplies them by the pointwise
// weights of the slab's 32 channels (4 K-steps), and stores affine + ReLU + 16-bit rounding of the result into the halo slab --
// exactly the values the separate layer would have written to memory (same K order, same rounding), zeros outside the image.
// The intermediate tensor never exists: -2 x 51 MB of traffic and one launch per C2 step for +3 % matrix work in this kernel.
typedef short s16x8_pool __attribute__((ext_vector_type(8)));
#define NTAPS_EVEN_STEPS(KD, KH, KW, TPS) (((taps_padded((KD) * (KH) * (KW)) / (TPS)) & 1) == 0)

// PERSIST (round 6; two-phase form, one channel group: p.gy == 1): the workgroup is a PERSISTENT tile loop -- the grid is p.gpersist workgroups
// (one per CU) and workgroup w runs the virtual block ids w, w + gridDim.x, ... of the p.gcount ids the ordinary launch would have had (same
// XCD-aware remap: consecutive tiles of one workgroup stay on its XCD's share of the map).  What the loop buys over one workgroup per tile:
//   * no relaunch gap between a CU's tiles (measured 2.0 us of a 42 us conv3d_2c workgroup, tools/timeline_probe.py);
//   * the weight ring never drains: the last three steps of a tile request the first three steps' weights of the next one (the same
//     weights: one channel group), so a tile starts with ring buffer 0 filled and both register sets in flight -- the prologue's weight
//     round trip (0.7-1.2 us) and the scale / shift fetch happen once per workgroup;
//   * the next tile's halo is REQUESTED before the current tile's epilogue (index tables + global loads, raw pixels of the fused pointwise
//     input for PRE) and lands while the epilogue's conversions, LDS passes and stores run; the fragment registers of the K loop are dead there.
// The accumulation order of every output is unchanged: bit-identical to the one-tile-per-workgroup launch.
template <typename T, int TWL, int NB, int KD, int KH, int KW, int TPS, int MB, int WV, int PH = 0, bool GRP = false, bool PRE = false, bool POOL = false, bool PERSIST = false>
__device__ __forceinline__ void conv_tap_body(const ConvParams& p) {
    static_assert(!PERSIST || (PH == 1 && !GRP && (NTAPS_EVEN_STEPS(KD, KH, KW, TPS))), "the persistent tile loop exists for the two-phase form with an even number of steps per slab");
    static_assert(!(PERSIST && POOL) || PRE, "persistent + pooled epilogue: the pooled tile lives in halo + stash (PRE)");
    static_assert(!POOL || (TWL == 3 && WV == 8 && PH == 1 && !GRP && sizeof(T) == 2 && KD == 3), "the pooled epilogue exists for the 4-plane 8x8 tile of the two-phase 16-bit form");
    static_assert(MB == 2 && (WV == 8 || WV == 4), "two accumulator rows per wave; 8 or 4 waves");
    static_assert(!PRE || (PH == 1 && sizeof(T) == 2), "the fused pointwise input exists for the two-phase 16-bit form");
    static_assert(PH == 0 || (WV == 8 && TPS == 2 && sizeof(T) == 2), "two-phase form: 8 waves, two taps per step, 16-bit storage");
    constexpr int NT = WV * 64;                 // threads
    constexpr int WM = WV / 2;                  // waves along the pixel axis (x 2 along channels)
    constexpr int TPXM = WM * MB * 32;          // pixels of a full tile: 256 (WV = 8) or 128 (WV = 4)
    // tile shapes: TWL = 4 -> 1 plane x 16 x 16, TWL = 5 -> 1 x 8 x 32, TWL = 3 -> 4 planes x 8 x 8 (small maps:
    // 7x7 ROI features would fill 19 % of a 16x16 tile; four planes of 8x8 fill 77 %), TWL = 0 -> a box of
    // p.gtd x p.gth x p.gtw <= 256 pixels chosen at launch (GEN): power-of-two tiles cover a 28x28 map at 77 %,
    // a 50x50 one at 70 %; 2x4x28 and 2x5x25 boxes reach 88 % and 98 %.  The index arithmetic of a GEN tile uses
    // divisions, but only outside the step loop.
    // (WV = 4 halves the tile along its leading axis: 8 x 16, 4 x 32, 2 planes x 8 x 8, or a general box <= 128 pixels)
    constexpr bool GEN = (TWL == 0);
    constexpr int TDL = (TWL == 3) ? (TPXM == 256 ? 2 : 1) : 0;
    constexpr int THC = GEN ? 1 : (TPXM >> (TWL + TDL));                     // tile rows of a power-of-two tile: 16, 8 or 4
    constexpr int THL = GEN ? 0 : (THC == 16 ? 4 : (THC == 8 ? 3 : 2));      // log2(TH)
    const int TD = GEN ? p.gtd : (1 << TDL);
    const int TW = GEN ? p.gtw : (1 << TWL), TH = GEN ? p.gth : THC;
    const int TPX = TD * TH * TW;                           // pixels of the tile (TPXM unless GEN)
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
    constexpr int NPIX_MAX = GEN ? conv_gen_npix(WV, NB) : ((1 << TDL) + KD - 1) * (THC + KH - 1) * ((1 << TWL) + KW - 1 + HWPAD);
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
    // LDS pitch of one tap's weight tile.  When the tile is not a whole number of 512-thread rounds (NB = 3: 768 vectors)
    // its last round is predicated -- a handful of exec-masked blocks in the step loop; if LDS has room the tile is
    // padded to whole rounds instead and every thread stores unconditionally (its surplus vectors are never read).
    constexpr bool PADB = BVEC > NT && (BVEC % NT) != 0 && (NPIX_MAX * PITCH + 3 * TPS * Q * NT * 16) <= 160 * 1024;   // (a tile below one round, NB = 1, is a wave-uniform predicate)
    constexpr int BPITCH = PADB ? Q * NT * 16 : BTILE;
    constexpr int BSTEP = TPS * BPITCH;                     // LDS weight bytes per step
    typedef typename Ld16<T>::type vec16;
    typedef typename frag<T>::type frag_t;

    // 16-bit storage: the accumulators are kept TRANSPOSED (TR: the MFMA takes the weight fragment as its first operand, so a lane
    // owns ONE pixel and, per register quad, four consecutive channels) -- the epilogue then needs no LDS transpose, see below
    constexpr bool TR = (ES == 2);
    constexpr int SS_BYTES = TR ? NBT * 32 * 2 * 4 : 0;                       // fp32 scale | shift of the tile's channels
    constexpr int BBYTES = PH ? 2 * 2 * TPS * (NB * KS * FRAGB) : 3 * BSTEP;    // two-phase form: two groups x a ring of two half-step buffers
    constexpr int STASH_BYTES = PRE ? NPIX_MAX * 64 : 0;                      // PRE: the second slab's 32 channels wait here (dense 64-byte pixels)
    constexpr int LDS_BYTES = NPIX_MAX * PITCH + BBYTES + SS_BYTES + STASH_BYTES;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
    // layout: halo | weight rings | scale table | stash.  PERSIST: halo | stash | scale table | weight rings -- the pooled epilogue's tile then
    // covers halo + stash only and the rings (which already hold the NEXT tile's first step) survive it
    static_assert(!POOL || PERSIST || TPXM * (NBT * 64 + 16) <= NPIX_MAX * PITCH + BBYTES, "the pooled epilogue's tile lives in the halo / weight rings, below the scale table");
    static_assert(!POOL || !PERSIST || TPXM * (NBT * 64 + 16) <= NPIX_MAX * PITCH + STASH_BYTES, "persistent form: the pooled tile lives in halo + stash");
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
    unsigned char* const ldsA = lds;
    unsigned char* const ldsB = lds + NPIX_MAX * PITCH + (PERSIST ? STASH_BYTES + SS_BYTES : 0);
    float* const ldsS = (float*)(lds + NPIX_MAX * PITCH + (PERSIST ? STASH_BYTES : BBYTES));
    unsigned char* const ldsP = lds + NPIX_MAX * PITCH + (PERSIST ? 0 : BBYTES + SS_BYTES);
    // tile pixel index m (accumulator row) -> box coordinates, and whether the row holds a pixel of the box at all.
    // General boxes, linear mode: rows past the box alias pixel 0.  General boxes, p.gmode = 1 (box widths just below a multiple
    // of 16: the 14- and 28-wide C2 maps, 13): the 16 lanes of every ds_read_b128 service group ({0-3,12-15,20-27} and
    // {4-11,16-19,28-31} of a 32-row block) take 16 CONSECUTIVE columns of ONE box row -- consecutive halo indices, so the group is
    // conflict-free whatever the row pitch (linear packing of 14-wide rows measured 27-34 % of the LDS cycles as bank conflicts);
    // the lanes past the row's end (2 of 16 at width 14, 4 of 32 at width 28: no more padding than the linear packing of
    // these boxes leaves anyway) keep walking the halo row -- in-bounds reads whose products are never stored.
    auto tile_pix = [&](int m, int& td, int& th, int& tw) -> bool {
        if (GEN) {
            if (p.gmode) {
                const int pl_ = m & 31, blk = m >> 5;
                // service group and position inside it: 0-3 -> (0, 0..3), 4-11 -> (1, 0..7), 12-15 -> (0, 4..7), 16-19 -> (1, 8..11),
                // 20-27 -> (0, 8..15), 28-31 -> (1, 12..15)
                int g, j;
                if (pl_ < 4) { g = 0; j = pl_; }
                else if (pl_ < 12) { g = 1; j = pl_ - 4; }
                else if (pl_ < 16) { g = 0; j = pl_ - 8; }
                else if (pl_ < 20) { g = 1; j = pl_ - 8; }
                else if (pl_ < 28) { g = 0; j = pl_ - 12; }
                else { g = 1; j = pl_ - 16; }
                const int spr = (TW + 15) >> 4;                       // 16-column runs per box row
                const int slot = blk * 2 + g;
                const int row = slot / spr, col = (slot % spr) * 16 + j;
                const bool ok = row < TD * TH && col < TW;
                const int rc = row < TD * TH ? row : 0;
                tw = col; th = rc % TH; td = rc / TH;
                return ok;
            }
            const int mc = m < TPX ? m : 0;
            tw = mc % TW; const int q = mc / TW; th = q % TH; td = q / TH;
            return m < TPX;
        } else {
            td = m >> (TWL + THL); th = (m >> TWL) & (TH - 1); tw = m & (TW - 1);
            return true;
        }
    };

    auto tile_col = [](int th, int j) {                    // tile row th, accumulator-row column slot j -> tile column
        if (TWL == 4) return (th & 1) ? ((j + 14) & 15) : j;
        if (TWL == 3) return (((th & 3) == 1) || ((th & 3) == 2)) ? (j ^ 4) : j;
        return j;
    };
    // (not const: the persistent tile loop re-derives them from an opaque copy of the thread index once per tile -- see relaunder())
    int tid = threadIdx.x;
    int lane = tid & 63;
#ifdef STEP_EMUL
    int wave = tid >> 6;
#else
    int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform -> scalar branches
#endif
    int khalf = lane >> 5;
    int wm = (PH == 2) ? (wave >> 1) : (wave % WM), wn = (PH == 2) ? (wave & 1) : (wave / WM);
    // PERSIST: everything the epilogue and the halo staging derive from the thread index (LDS / global addresses, pixel tables, pooling
    // item decode: ~150 values) is the same for every tile, so the compiler hoists it out of the tile loop and keeps it alive across the K
    // loop, whose 222 VGPRs leave no room: 133 spilled registers, and every scratch reload is an s_waitcnt vmcnt(0) that also drains the
    // halo requests in flight.  Passing the index through an empty asm statement once per tile makes those values per-tile values again
    // (what a freshly launched workgroup computes anyway); the K loop's own tables (aoff, wthr, bwave) were derived before and stay put.
    auto relaunder = [&]() {
#ifndef STEP_EMUL
        asm volatile("" : "+v"(tid));
        asm volatile("" : "+s"(wave));
#endif
        lane = tid & 63; khalf = lane >> 5;
        wm = (PH == 2) ? (wave >> 1) : (wave % WM); wn = (PH == 2) ? (wave & 1) : (wave / WM);
    };

    int gbx, gby;
    unsigned vid = blockIdx.x - (unsigned)p.gbase;          // PERSIST: the virtual block id this workgroup is working on
    if constexpr (PERSIST) {
        while (vid < (unsigned)p.gcount && !grid_coords_of(p, vid, gbx, gby)) vid += gridDim.x;     // (padding ids of the remapped grid)
        if (vid >= (unsigned)p.gcount) return;
    } else {
        if (!grid_coords(p, gbx, gby)) return;
    }
    // (Measured and removed: starting the first round's workgroups a pseudo-random 0..27 us apart -- to de-phase the CUs, whose
    // equal-length tiles bring every epilogue's store burst to the same moment -- only ADDS the delay: conv3d_2c 238.7 us ->
    // 242 / 240 / 243 / 248 / 270 us at 64 ... 1024 x 64 clocks of spread (gpurun_out/ab_desync.log, round 3).  The epilogues do
    // not contend with each other.)
    STEP_PROBE_IDS(p);
    STEP_PROBE_MARK(p, 0);
    struct TileC { int tw_i, th_i, d0, n, h0, w0; };
    auto tile_of = [&](int bx) {
        TileC c;
        int t = bx + p.tile0;
        c.tw_i = t % p.tiles_w; t /= p.tiles_w;
        c.th_i = t % p.tiles_h; t /= p.tiles_h;
        c.d0 = (t % p.tiles_d) * TD;
        c.n = t / p.tiles_d;
        c.h0 = c.th_i * TH; c.w0 = c.tw_i * TW;
        return c;
    };
    TileC tc = tile_of(gbx);                                 // the tile being computed (PERSIST: replaced at the end of the tile loop's body)
    int tw_i = tc.tw_i, th_i = tc.th_i, d0 = tc.d0, n = tc.n, h0 = tc.h0, w0 = tc.w0;
    const int HHW = HH_ * HW_;
    const int nb0 = gby * NBT;
    const int KC16 = p.nchunks32 * 2;
    const int nslab = (p.Cin + CKT - 1) / CKT;
    const int S = nslab * SPS;            // pipeline steps

    const T* xg = (const T*)p.x;
    const unsigned char* wg = (const unsigned char*)p.w;

    // LDS address of this lane's two accumulator rows (before the tap shift), incl. its k half
    const unsigned char* abase[MB];          // (not const-qualified pointers: the two-phase form launders them per slab)
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

    // Per-thread staging table (the same for every slab): element offset of each of this thread's halo vectors, or
    // ~0u outside the image / the halo.  Decoding a vector index into (plane, row, column) takes divisions -- by
    // run-time box dimensions in the general-tile instantiation -- which used to be redone for every slab.
    // (32-bit offsets: the planner sends tensors of >= 2^32 elements to conv_igemm_kernel.)
    unsigned goff[ITER];
    auto build_goff = [&](const TileC& c) {
    const int d0 = c.d0, h0 = c.h0, w0 = c.w0, n = c.n;
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int v = tid + it * NT;
        goff[it] = ~0u;
        if (v < NVEC) {
            const int pix = v / SLOTS, slot = v % SLOTS;
            int plane, rem, r, cc;
            if constexpr (GEN) {                                   // run-time extents: multiply by the host's reciprocals (p.mag_*) instead of dividing
                plane = (int)__umulhi((unsigned)pix, p.mag_hhw); rem = pix - plane * HHW;
                r = (int)__umulhi((unsigned)rem, p.mag_hw); cc = rem - r * HW_;
            } else { plane = pix / HHW; rem = pix % HHW; r = rem / HW_; cc = rem % HW_; }
            const int id = d0 + plane - KD / 2, ih = h0 + r - KH / 2, iw = w0 + cc - KW / 2;
            const bool inb = id >= 0 && id < p.D && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W && cc < HWV;
            if (inb) {
                const size_t gpix = (((size_t)n * p.D + id) * p.H + ih) * p.W + iw;
                goff[it] = (unsigned)(gpix * p.x_cstride + p.x_coff + slot * VEC);
            }
        }
    }
    };
    if constexpr (PH == 0) build_goff(tc);                 // (the two-phase form requests its first weights before this index work)
    // the tile's scale / shift: requested with the first loads, parked in LDS behind the halo stores (visible after the prologue's
    // barrier; read in the epilogue)
    float ss_sc = 1.f, ss_sh = 0.f;
    auto ss_load = [&]() {
        if (TR && tid < NBT * 32) {
            const int co = min(nb0 * 32 + tid, p.Cout - 1);
            if (p.scale) ss_sc = p.scale[co];
            if (p.shift) ss_sh = p.shift[co];
        }
    };
    auto ss_store = [&]() {
        if (TR && tid < NBT * 32) { ldsS[tid] = ss_sc; ldsS[NBT * 32 + tid] = ss_sh; }
    };
    vec16 stage[ITER];
    auto stage_A = [&](int slab, int part = 3) {             // part 1: global -> registers, 2: registers -> LDS, 3: both
        if (part & 1) {
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int v = tid + it * NT;
            vec16 val;
#pragma unroll
            for (int e = 0; e < VEC; ++e) val[e] = 0;
            const int c = slab * CKT + (v % SLOTS) * VEC;
            if (goff[it] != ~0u && c < p.Cin) val = *(const vec16*)(xg + (size_t)goff[it] + slab * CKT);
            stage[it] = val;
        }
        }
        if (part & 2) {
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int v = tid + it * NT;
            if (v < NVEC) {
                const int pix = v / SLOTS, slot = v % SLOTS;
                *(vec16*)(ldsA + pix * PITCH + (slot << 4)) = stage[it];
            }
        }
        }
    };

    // ---- PRE: per-wave table of the halo's 32-pixel blocks (block b = wave + bi * WV): element offset of lane's pixel or ~0u
    constexpr int PBLK = (NPIX_MAX + 31) / 32, PBW = (PBLK + WV - 1) / WV;
    unsigned poff[PRE ? PBW : 1];
    auto build_poff = [&](const TileC& c) {
        const int d0 = c.d0, h0 = c.h0, w0 = c.w0, n = c.n;
#pragma unroll
        for (int bi = 0; bi < (PRE ? PBW : 0); ++bi) {
            const int pix = (wave + bi * WV) * 32 + (lane & 31);
            poff[bi] = ~0u;
            if (pix < NPIX) {
                int plane, rem, r, cc;
                if constexpr (GEN) {
                    plane = (int)__umulhi((unsigned)pix, p.mag_hhw); rem = pix - plane * HHW;
                    r = (int)__umulhi((unsigned)rem, p.mag_hw); cc = rem - r * HW_;
                } else { plane = pix / HHW; rem = pix % HHW; r = rem / HW_; cc = rem % HW_; }
                const int id = d0 + plane - KD / 2, ih = h0 + r - KH / 2, iw = w0 + cc - KW / 2;
                if (id >= 0 && id < p.D && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W && cc < HWV)
                    poff[bi] = (unsigned)(((((size_t)n * p.D + id) * p.H + ih) * p.W + iw) * p.x_cstride + p.x_coff);
            }
        }
    };
    // slab 0: ONE read of the raw pixels (operand-layout loads touch 32 cache lines per instruction: the expensive part) feeds both
    // 32-channel slabs -- slab 0's values go to the halo slab, slab 1's wait in ldsP; slab 1: an LDS -> LDS copy
    constexpr int PKS = 4;                                                   // K-steps of the pointwise layer: 64 input channels
    frag_t fap[PRE ? PBW : 1][PKS];                                          // the raw pixels of the halo's blocks, MFMA operand layout
    auto pre_issue = [&]() {                                                 // every request first: one memory round trip
        if constexpr (PRE) {
#pragma unroll
            for (int bi = 0; bi < PBW; ++bi) {
                const T* src = xg + (poff[bi] != ~0u ? poff[bi] : 0u) + 8 * khalf;
#pragma unroll
                for (int s = 0; s < PKS; ++s) fap[bi][s] = *(const frag_t*)(src + 16 * s);
            }
        }
    };
    auto stage_pre = [&](int slab, bool issue = true) {
        if constexpr (PRE) {
            if (slab != 0) {
#pragma unroll 1
                for (int v = tid; v < NPIX * 4; v += NT)
                    *(u32x4*)(ldsA + (v >> 2) * PITCH + (v & 3) * 16) = *(const u32x4*)(ldsP + v * 16);
                return;
            }
            if (issue) pre_issue();                                          // (PERSIST: requested before the previous tile's epilogue)
#pragma unroll
            for (int nbk = 0; nbk < 2; ++nbk) {
                const unsigned char* pw = (const unsigned char*)p.pre_w + ((size_t)nbk * PKS) * FRAGB + lane * 16;   // n-block nbk (taps_padded(1) = 1)
                frag_t fbp[PKS];
#pragma unroll
                for (int s = 0; s < PKS; ++s) fbp[s] = *(const frag_t*)(pw + s * FRAGB);
                f32x4 sc[4], sh[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 one = {1.f, 1.f, 1.f, 1.f}, zero = {0.f, 0.f, 0.f, 0.f};
                    sc[g] = p.pre_scale ? *(const f32x4*)(p.pre_scale + nbk * 32 + 8 * g + 4 * khalf) : one;
                    sh[g] = p.pre_shift ? *(const f32x4*)(p.pre_shift + nbk * 32 + 8 * g + 4 * khalf) : zero;
                }
#pragma unroll
                for (int bi = 0; bi < PBW; ++bi) {
                    const int pix = (wave + bi * WV) * 32 + (lane & 31);
                    if ((wave + bi * WV) * 32 >= NPIX) continue;             // (wave-uniform)
                    f32x16 a;
#pragma unroll
                    for (int r = 0; r < 16; ++r) a[r] = 0.f;
#pragma unroll
                    for (int s = 0; s < PKS; ++s) mma_k16(fbp[s], fap[bi][s], a, T());   // weights first: a lane owns its pixel's channels 8g + 4 khalf + {0..3}
                    const bool inb = poff[bi] != ~0u;
                    unsigned d[4][2];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v[e] = fmaxf(a[4 * g + e] * sc[g][e] + sh[g][e], 0.f); v[e] = inb ? v[e] : 0.f; }
                        d[g][0] = (unsigned)elem<T>::bits16(v[0]) | ((unsigned)elem<T>::bits16(v[1]) << 16);
                        d[g][1] = (unsigned)elem<T>::bits16(v[2]) | ((unsigned)elem<T>::bits16(v[3]) << 16);
                    }
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        lane32_swap(d[2 * h][0], d[2 * h + 1][0]);
                        lane32_swap(d[2 * h][1], d[2 * h + 1][1]);
                        if (pix < NPIX) {
                            const u32x4 o = {d[2 * h][0], d[2 * h][1], d[2 * h + 1][0], d[2 * h + 1][1]};
                            unsigned char* dst = nbk == 0 ? ldsA + pix * PITCH : ldsP + pix * 64;
                            *(u32x4*)(dst + (16 * h + 8 * khalf) * ES) = o;
                        }
                    }
                }
            }
        }
    };

    // the tile's epilogue (reads acc and the current tile's coordinates n, d0, h0, w0, th_i, tw_i; PERSIST calls it once per tile)
    // (mid(): called once when the accumulators are dead or about to be -- POOL: behind the barrier that follows the tile's trip to LDS, otherwise
    // at the start; the persistent loop requests the next tile's halo there)
    auto epilogue = [&](auto&& mid) {
    T* yg = (T*)p.y;
    if constexpr (!POOL) mid();
    const T* rg = (const T*)p.res;
    if constexpr (TR) {
        // 16-bit outputs, transposed accumulators: lane l owns pixel (l & 31) of each of its MB row blocks and, in registers
        // 4g .. 4g+3 of a 32x32 tile, the four consecutive channels 8g + 4 (l >> 5) + {0..3}.  After the affine / residual / ReLU
        // the four values are two packed dwords; ONE v_permlane32_swap per dword pair (g even, g odd) gives the lower lane channels
        // 8g' .. 8g'+7 and the upper lane 8g'+8 .. 8g'+15 of the same pixel: every lane stores 16 contiguous bytes straight from
        // registers.  No LDS transpose, no barrier (the LDS form took 5.3 of conv3d_2c's 44 us per tile: two passes of 48
        // ds_write_b32 + barrier + read-out, tools/timeline_probe.py).
        long long opix[MB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            int tdl, thl, twl;
            const bool inbox = tile_pix(wm * (MB * 32) + mb * 32 + (lane & 31), tdl, thl, twl);
            const int od = d0 + tdl, oh = h0 + thl, ow = w0 + tile_col(thl, twl);
            opix[mb] = (inbox && od < p.D && oh < p.H && ow < p.W) ? (((long long)n * p.D + od) * p.H + oh) * p.W + ow : -1;
        }
        if constexpr (POOL) {
            // POOL (step_conv_forward_pre_pool: maxPool3d_3a -- (1,3,3) / (1,2,2), TF padding (0,1) -- taken on conv3d_2c's tile while it is
            // on the chip): y is the POOLED tensor [N, D, Hp, Wp, C]; the un-pooled output never exists.  Same scheme as the stem's pooled
            // epilogue (stem.hip): the 4 planes x 8 x 8 tile goes to LDS pixel-major (the rings are free: the K loop ended in a barrier); a
            // pooled pixel (ph, pw) is the max over rows 2ph .. 2ph+2 and columns 2pw .. 2pw+2, so a tile plane holds everything for 3 of its
            // 4 pooled rows / columns and two of the three rows / columns of the fourth.  The tile writes the max over what it HAS to y and
            // its own first row and first column (raw values) to pool_row / pool_col; pool_seam_fix_kernel completes the pooled pixels on tile
            // seams from those.  Values are post-ReLU (>= +0): the zero padding of the reference's ConstantPad3d is neutral, out-of-image
            // pixels of partial tiles enter as 0, and 16-bit patterns order like signed integers (one v_pk_max_i16 per pair).
            constexpr int TP = NBT * 64 + 16;                            // bytes per tile pixel (+16: 16 lanes x 16 B at one channel offset spread over all banks)
            int tpix[MB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                int tdl, thl, twl;
                tile_pix(wm * (MB * 32) + mb * 32 + (lane & 31), tdl, thl, twl);
                tpix[mb] = (tdl * 8 + thl) * 8 + tile_col(thl, twl);
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int cl = (wn * NB + i) * 32;
                f32x4 sc[4], sh[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    sc[g] = *(const f32x4*)(ldsS + cl + 8 * g + 4 * khalf);
                    sh[g] = *(const f32x4*)(ldsS + NBT * 32 + cl + 8 * g + 4 * khalf);
                }
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    const bool okp = opix[mb] >= 0;
                    unsigned d[4][2];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(acc[mb][i][4 * g + e] * sc[g][e] + sh[g][e], 0.f);      // (the host admits relu = 1 only)
                        d[g][0] = (unsigned)elem<T>::bits16(v[0]) | ((unsigned)elem<T>::bits16(v[1]) << 16);
                        d[g][1] = (unsigned)elem<T>::bits16(v[2]) | ((unsigned)elem<T>::bits16(v[3]) << 16);
                    }
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        lane32_swap(d[2 * h][0], d[2 * h + 1][0]);
                        lane32_swap(d[2 * h][1], d[2 * h + 1][1]);
                        u32x4 o = {d[2 * h][0], d[2 * h][1], d[2 * h + 1][0], d[2 * h + 1][1]};
                        if (!okp) o = u32x4{0u, 0u, 0u, 0u};
                        *(u32x4*)(lds + tpix[mb] * TP + (cl + 16 * h + 8 * khalf) * 2) = o;
                    }
                }
            }
            __syncthreads();
            mid();
            constexpr int CV = NBT * 4;                                  // 8-channel vectors of the workgroup's channels
            unsigned short* yp_ = (unsigned short*)p.y;
            // 4 planes x 4 x 4 pooled pixels x CV vectors = NB items per thread; vector fastest (a pixel's channels are contiguous)
            // (PERSIST: one item at a time -- unrolled, the scheduler hoists all 9 * NB window reads to the top, ~110 live registers beside the
            // halo requests mid() has just put in flight, and the allocator spills those)
#pragma unroll PERSIST ? 1 : NB
            for (int k = 0; k < NB; ++k) {
                const int item = tid + NT * k;
                const int v = item % CV, pq = item / CV;
                const int j = pq & 3, i2 = (pq >> 2) & 3, pl_ = pq >> 4;
                const unsigned char* base = lds + ((pl_ * 8 + 2 * i2) * 8 + 2 * j) * TP + v * 16;
                u32x4 m = *(const u32x4*)base;
#pragma unroll
                for (int dr = 0; dr < 3; ++dr)
#pragma unroll
                    for (int dc = 0; dc < 3; ++dc) {
                        if (dr == 0 && dc == 0) continue;
                        if (2 * i2 + dr < 8 && 2 * j + dc < 8) {
                            const u32x4 o = *(const u32x4*)(base + (dr * 8 + dc) * TP);
                            m = __builtin_bit_cast(u32x4, __builtin_elementwise_max(__builtin_bit_cast(s16x8_pool, m), __builtin_bit_cast(s16x8_pool, o)));
                        }
                    }
                const int od = d0 + pl_, ph = (h0 >> 1) + i2, pw = (w0 >> 1) + j, co = nb0 * 32 + v * 8;
                if (od < p.D && ph < p.Hp && pw < p.Wp && co < p.Cout)
                    *(u32x4*)(yp_ + ((((size_t)n * p.D + od) * p.Hp + ph) * p.Wp + pw) * p.y_cstride + p.y_coff + co) = m;
            }
            // the tile's first row -> pool_row, first column -> pool_col (raw values; tiles of the first tile row / column have no reader)
            unsigned short* rb = (unsigned short*)p.pool_row;
            unsigned short* cb = (unsigned short*)p.pool_col;
#pragma unroll PERSIST ? 1 : NB
            for (int k = 0; k < NB; ++k) {
                const int item = tid + NT * k;                           // 2 x 4 planes x 8 pixels x CV vectors = NB x 512
                const int v = item % CV, q = item / CV;
                const int e = q & 7, pl_ = (q >> 3) & 3, col_item = q >> 5;
                const int od = d0 + pl_, co = nb0 * 32 + v * 8;
                if (od >= p.D || co >= p.Cout) continue;
                const size_t plane = (size_t)n * p.D + od;
                if (!col_item) {
                    if (th_i > 0 && w0 + e < p.W)
                        *(u32x4*)(rb + ((plane * p.tiles_h + th_i) * p.W + w0 + e) * (size_t)p.Cout + co) = *(const u32x4*)(lds + ((pl_ * 8) * 8 + e) * TP + v * 16);
                } else {
                    if (tw_i > 0 && h0 + e < p.H)
                        *(u32x4*)(cb + ((plane * p.tiles_w + tw_i) * p.H + h0 + e) * (size_t)p.Cout + co) = *(const u32x4*)(lds + ((pl_ * 8 + e) * 8) * TP + v * 16);
                }
            }
#ifdef STEP_PROBE
            if constexpr (!PERSIST) {
                STEP_PROBE_MARK(p, 3);
                __builtin_amdgcn_s_waitcnt(0);
                probe_clock_end(p.probe);
                STEP_PROBE_MARK(p, 4);
            }
#endif
            return;
        }
        if (p.vec_epi) {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int cl = (wn * NB + i) * 32;                       // first channel of the block inside the workgroup tile
                f32x4 sc[4], sh[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    sc[g] = *(const f32x4*)(ldsS + cl + 8 * g + 4 * khalf);
                    sh[g] = *(const f32x4*)(ldsS + NBT * 32 + cl + 8 * g + 4 * khalf);
                }
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    const bool okp = opix[mb] >= 0;
                    const size_t obase = (size_t)(okp ? opix[mb] : 0);
                    unsigned d[4][2];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[mb][i][4 * g + e] * sc[g][e] + sh[g][e];
                        const int co = nb0 * 32 + cl + 8 * g + 4 * khalf;
                        if (rg && okp && co < p.Cout) {
                            const u16x4 rv = *(const u16x4*)(rg + obase * p.r_cstride + p.r_coff + co);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] += elem<T>::from_bits16(rv[e]);
                        }
                        if (p.relu) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                        }
                        d[g][0] = (unsigned)elem<T>::bits16(v[0]) | ((unsigned)elem<T>::bits16(v[1]) << 16);
                        d[g][1] = (unsigned)elem<T>::bits16(v[2]) | ((unsigned)elem<T>::bits16(v[3]) << 16);
                    }
#pragma unroll
                    for (int h = 0; h < 2; ++h) {                        // register quads (2h, 2h + 1) -> one 16-byte run per lane
                        lane32_swap(d[2 * h][0], d[2 * h + 1][0]);
                        lane32_swap(d[2 * h][1], d[2 * h + 1][1]);
                        const int co = nb0 * 32 + cl + 16 * h + 8 * khalf;
                        if (okp && co < p.Cout) {
                            const u32x4 o = {d[2 * h][0], d[2 * h][1], d[2 * h + 1][0], d[2 * h + 1][1]};
                            *(u32x4*)(yg + obase * p.y_cstride + p.y_coff + co) = o;
                        }
                    }
                }
            }
#ifdef STEP_PROBE
            STEP_PROBE_MARK(p, 3);
            __builtin_amdgcn_s_waitcnt(0);                    // every store acknowledged
            probe_clock_end(p.probe);
            STEP_PROBE_MARK(p, 4);
#endif
            return;
        }
        // channel counts / offsets off the 16-byte grid: element stores (same transposed ownership)
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int cl = (wn * NB + i) * 32;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = nb0 * 32 + cl + cd_row(r, lane);
                    if (opix[mb] >= 0 && co < p.Cout) {
                        const size_t o = (size_t)opix[mb];
                        float v = acc[mb][i][r] * ldsS[cl + cd_row(r, lane)] + ldsS[NBT * 32 + cl + cd_row(r, lane)];
                        if (rg) v += elem<T>::to_f32(rg[o * p.r_cstride + p.r_coff + co]);
                        if (p.relu) v = fmaxf(v, 0.f);
                        yg[o * p.y_cstride + p.y_coff + co] = elem<T>::from_f32(v);
                    }
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
                    const bool inbox = tile_pix(mm, tdl, thl, twl);
                    const int od = d0 + tdl, oh = h0 + thl, ow = w0 + tile_col(thl, twl);
                    if (inbox && od < p.D && oh < p.H && ow < p.W) {
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
    };

    if constexpr (PH != 0) {
        // ---- two-phase pipeline (see the header comment).  A slab's steps are unrolled: tap shifts and ring-buffer offsets
        // are immediates of the ds_reads (measured: the scalar / branch code of the rolled loop cost ~290 cycles per phase,
        // as much as the 20 fragment reads themselves).
        constexpr int HB = NB * KS * FRAGB;              // bytes of one tap's weights for ONE channel half
        constexpr int HSTEP = TPS * HB;                  // ... of one step
        constexpr int QH = HSTEP / 16 / 256;             // 16-byte vectors per thread per step (256 threads per group) = NB
        static_assert(QH * 256 * 16 == HSTEP, "a step's half tile is a whole number of 256-thread rounds");
        const int grp = wn;                              // wave group = channel half
        // grouped launches (GRP): a wave group whose channel blocks all lie past Cout -- a narrow member that takes its instantiation
        // from a deeper partner, e.g. 48 channels at NB = 2 -- skips its fragment reads, weight stream and MFMAs and only keeps the
        // barriers and its share of the halo staging.  (Plain launches multiply a duplicate block instead: the uniform branches
        // cost 0.5-2 % on every layer, measured, and only odd block counts would gain.)
        const bool act = GRP ? (nb0 + grp * NB < p.nblk32) : true;
        const int gtid = (PH == 2) ? ((wave >> 1) * 64 + lane) : (tid & 255);     // thread index inside the group
        unsigned char* const ldsBg = ldsB + grp * (2 * HSTEP);                     // per group: a ring of TWO step buffers
        const unsigned char* wthr[QH];
#pragma unroll
        for (int q = 0; q < QH; ++q) {
            const int v = gtid + q * 256;
            const int tp = v / (HB / 16), rem = v % (HB / 16);
            const int f = rem / FRAGV, within = rem % FRAGV;
            const int nbl = f / KS, ks = f % KS;
            const int nbg = min(nb0 + grp * NB + nbl, p.nblk32 - 1);
            wthr[q] = wg + (((size_t)nbg * taps_padded(NTAPS) + tp) * KC16 + ks) * FRAGB + within * 16;
        }
        const unsigned wstep = (unsigned)TPS * KC16 * FRAGB;               // bytes between consecutive steps of a slab
        auto woff_of = [&](int slab, int si) -> unsigned {                  // weight offset of a step; past the end: the last one (PERSIST: the next tile's first steps)
            if (slab >= nslab) { if constexpr (PERSIST) slab -= nslab; else { slab = nslab - 1; si = SPS - 1; } }
            return (unsigned)slab * (KS * FRAGB) + (unsigned)si * wstep;
        };
        auto load_B = [&](unsigned woff, u32x4 (&r)[QH]) {
#pragma unroll
            for (int q = 0; q < QH; ++q) r[q] = *(const u32x4*)(wthr[q] + (size_t)woff);
        };
        auto store_B = [&](int bufoff, const u32x4 (&r)[QH]) {
#pragma unroll
            for (int q = 0; q < QH; ++q) *(u32x4*)(ldsBg + bufoff + (gtid + q * 256) * 16) = r[q];
        };
        u32x4 R0[QH], R1[QH];                                // weights in flight: R0 even steps, R1 odd steps
        frag_t fa[TPS][KS][MB], fb[TPS][KS][NB];
        const unsigned char* const bwave = ldsBg + lane * (8 * ES);
        const int HHWB = HHW * PITCH, HWB = HW_ * PITCH;    // (compile-time constants for the power-of-two tiles)
        int aoff[MB];                                        // LDS byte offsets of this lane's accumulator rows
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) aoff[mb] = (int)(abase[mb] - lds);
        // step g (global index) of parity q: fragments from ring buffer q; register set q^1 holds step g+1, written to ring
        // buffer q^1 (last read in L of step g-1) and re-issued for step g+3
        auto step = [&](auto sic, auto parc, int slab) {
            constexpr int SI = decltype(sic)::value;               // step inside the slab
            constexpr int Q = (decltype(parc)::value + SI) & 1;    // parity of the global step
            u32x4 (&Rn)[QH] = Q ? R0 : R1;
            // ---- L
            if (act) {
#pragma unroll
                for (int tp = 0; tp < TPS; ++tp) {
                    const int tap = SI * TPS + tp;                 // (compile-time after unrolling)
                    // the zero tap that pads an odd tap count: no fragment reads and (below) no MFMAs -- its products are exact zeros,
                    // 1 / 28 of the loop's matrix work and LDS reads (round 6; the classic form still multiplies it)
#ifndef STEP_EXP_KEEP_PAD_TAP                                       // (experiment builds: make EXP=padtap EXPFLAGS=-DSTEP_EXP_KEEP_PAD_TAP, tools/step_ab.py)
                    if (tap >= NTAPS) continue;
#endif
                    int shift = tap >= NTAPS ? 0 : (tap / (KH * KW)) * HHWB + ((tap / KW) % KH) * HWB + (tap % KW) * PITCH;
#ifndef STEP_EMUL
                    // run-time tile boxes: the shift is not an immediate.  Pin its use to this phase (volatile asm statements
                    // keep their order with the barriers): left alone the compiler forms all 2 x 27 shifted fragment addresses
                    // at the top of the slab (54 live VGPRs -> scratch spills)
                    if (GEN) asm volatile("" : "+s"(shift));
#endif
#pragma unroll
                    for (int j = 0; j < KS; ++j) {
#pragma unroll
                        for (int mb = 0; mb < MB; ++mb) fa[tp][j][mb] = lds_read_bfrag<T>(lds + aoff[mb] + shift + j * 32);
#pragma unroll
                        for (int i = 0; i < NB; ++i) fb[tp][j][i] = lds_read_bfrag<T>(bwave + Q * HSTEP + tp * HB + (i * KS + j) * FRAGB);
                    }
                }
            }
            if (act) store_B((Q ^ 1) * HSTEP, Rn);
#ifndef STEP_EMUL
            __builtin_amdgcn_sched_barrier(0);             // the weight requests go BEHIND the fragment reads (see below)
#endif
            if (act) {
                constexpr int S3 = SI + 3;
                load_B(woff_of(slab + (S3 >= SPS ? 1 : 0), S3 >= SPS ? S3 - SPS : S3), Rn);
            }
#ifndef STEP_EMUL
            // keep the weight requests at the END of this phase: left to itself the compiler hoists them (into fresh registers)
            // to the head of the phase, where their issue delays the 20 ds_reads the partner wave's multiply phase is
            // supposed to cover (measured on conv3d_2c: 261 -> 320 us)
            __builtin_amdgcn_sched_barrier(0);
#endif
            __syncthreads();
#ifdef STEP_PROBE
            if (!PERSIST && slab == 0 && SI >= 1 && SI <= 4) STEP_PROBE_MARK(p, 7 + 2 * (SI - 1));          // slots 7, 9, 11, 13: L phase of step SI done (group 0)
#endif
            // ---- C: the step's MFMAs, back to back out of registers, at raised priority (the partner wave is in its L phase)
#ifndef STEP_EMUL
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#endif
            if (act) {
#pragma unroll
                for (int tp = 0; tp < TPS; ++tp) {
#ifndef STEP_EXP_KEEP_PAD_TAP
                    if (SI * TPS + tp >= NTAPS) continue;          // (the padding tap: see the L phase)
#endif
#pragma unroll
                    for (int j = 0; j < KS; ++j)
#pragma unroll
                        for (int i = 0; i < NB; ++i)
#pragma unroll
                            for (int mb = 0; mb < MB; ++mb) {
                                if constexpr (TR) mma_k16(fb[tp][j][i], fa[tp][j][mb], acc[mb][i], T());
                                else mma_k16(fa[tp][j][mb], fb[tp][j][i], acc[mb][i], T());
                            }
                }
            }
#ifndef STEP_EMUL
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
#endif
#ifdef STEP_PROBE
            if (!PERSIST && slab == 0 && SI >= 1 && SI <= 3) STEP_PROBE_MARK(p, 8 + 2 * (SI - 1));          // slots 8, 10, 12: MFMAs of step SI issued (before the barrier)
#endif
            __syncthreads();
        };
        // prologue: weights of steps 0 and 1 requested FIRST -- their addresses need nothing but the block index, and the ~400-900
        // instructions of index arithmetic behind the halo table (general boxes divide) then run under their latency instead of in
        // front of a second memory round trip (measured per workgroup: index tables 1.1-1.9 us, halo 0.5-1.6 us, weights + barrier
        // 0.7-1.2 us, one after the other) -- then halo slab 0; step 0 to ring buffer 0, steps 1 and 2 stay in the register sets
        if (act) {
            load_B(woff_of(0, 0), R0);
            load_B(woff_of(SPS > 1 ? 0 : 1, SPS > 1 ? 1 : 0), R1);
        }
#ifndef STEP_EMUL
        __builtin_amdgcn_sched_barrier(0);
#endif
        ss_load();
        if constexpr (PRE) build_poff(tc); else build_goff(tc);
        if constexpr (!PERSIST) STEP_PROBE_MARK(p, 5);
        if constexpr (PERSIST) {
            // the first tile's halo is only REQUESTED here; every tile's staging is finished at the top of the tile loop (one copy of that
            // code, and the accumulators -- cleared there -- are not alive while it runs)
            if constexpr (PRE) pre_issue(); else stage_A(0, 1);
        } else {
            if constexpr (PRE) stage_pre(0); else stage_A(0);
        }
        ss_store();
        if constexpr (!PERSIST) STEP_PROBE_MARK(p, 6);
        if (act) {
            store_B(0, R0);
            load_B(woff_of(SPS > 2 ? 0 : 1, SPS > 2 ? 2 : 2 - SPS), R0);
        }
        if constexpr (!PERSIST) __syncthreads();
        STEP_PROBE_MARK(p, 1);
        typedef std::integral_constant<int, 0> P0;
        typedef std::integral_constant<int, 1> P1;
#pragma unroll 1
        for (;;) {                                         // PERSIST: the workgroup's tile loop (otherwise a single pass)
        if constexpr (PERSIST) {
            if constexpr (PRE) stage_pre(0, false); else stage_A(0, 2);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int i = 0; i < NB; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mb][i][r] = 0.f;
            __syncthreads();
        }
#pragma unroll 1
        for (int slab = 0; slab < nslab; ++slab) {
            if (slab) {                                    // slab switch: realign the groups, re-stage the halo
                if (grp == 0) __syncthreads();            // (group 1's last C phase of the previous slab)
                if constexpr (PRE) stage_pre(slab); else stage_A(slab);
                __syncthreads();
            }
            if (grp == 1) __syncthreads();                // anti-phase: group 1 runs one phase behind group 0
            if ((SPS & 1) && (slab & 1)) static_for<SPS>([&](auto si) { step(si, P1(), slab); });
            else static_for<SPS>([&](auto si) { step(si, P0(), slab); });
        }
        if (grp == 0) __syncthreads();                    // realign before the epilogue reuses LDS
#ifdef STEP_PROBE
        int probe_tile = 0;                                // (probe build, PERSIST: slot 2 = first tile's K loop done, slots 5 .. 13 = end of tiles 0 .. 8, slot 4 = exit)
        if (!PERSIST || vid == blockIdx.x - (unsigned)p.gbase) STEP_PROBE_MARK(p, 2);
        if constexpr (PERSIST) probe_tile = (int)((vid - (blockIdx.x - (unsigned)p.gbase)) / gridDim.x);
#endif
        if constexpr (!PERSIST) {
            break;
        } else {
            // the workgroup's next tile.  Ring buffer 0 already holds its step 0 and the register sets its steps 1 and 2 (the last
            // steps' requests wrapped around: same channel group, same weights).  Its halo is requested from inside the epilogue (mid():
            // index table + global loads into registers the K loop's fragments occupied) and lands while the epilogue pools and stores.
            relaunder();
            unsigned nv = vid + gridDim.x;
            int nbx = 0, nby = 0;
            while (nv < (unsigned)p.gcount && !grid_coords_of(p, nv, nbx, nby)) nv += gridDim.x;
            const bool more = nv < (unsigned)p.gcount;
            TileC nt = tc;
            if (more) nt = tile_of(nbx);
            epilogue([&]() {
                if (more) {
                    if constexpr (PRE) { build_poff(nt); pre_issue(); } else { build_goff(nt); stage_A(0, 1); }
                } else {
                    // (no next tile: give the request registers a value all the same.  Left alone, "unchanged" means the PREVIOUS tile's
                    // requests stay alive from the top of the loop across the whole K loop to this point -- the allocator then spills them,
                    // and a spilled request is a load followed at once by s_waitcnt vmcnt(0) + scratch store: measured in the ISA, 4 of 12)
                    if constexpr (PRE) {
#pragma unroll
                        for (int bi = 0; bi < PBW; ++bi) {
                            poff[bi] = ~0u;
#pragma unroll
                            for (int s_ = 0; s_ < PKS; ++s_)
#pragma unroll
                                for (int e = 0; e < (int)(sizeof(frag_t) / sizeof(fap[0][0][0])); ++e) fap[bi][s_][e] = 0;
                        }
                    } else {
#pragma unroll
                        for (int it = 0; it < ITER; ++it) {
                            goff[it] = ~0u;
#pragma unroll
                            for (int e = 0; e < VEC; ++e) stage[it][e] = 0;
                        }
                    }
                }
            });
#ifdef STEP_PROBE
            if (probe_tile <= 8) STEP_PROBE_MARK(p, 5 + probe_tile);
            if (!more) { __builtin_amdgcn_s_waitcnt(0); probe_clock_end(p.probe); STEP_PROBE_MARK(p, 4); }
#endif
            if (!more) return;
            if constexpr (POOL) __syncthreads();          // (the pooled tile's readers are done: halo + stash may be overwritten)
            tc = nt; vid = nv;
            tw_i = tc.tw_i; th_i = tc.th_i; d0 = tc.d0; n = tc.n; h0 = tc.h0; w0 = tc.w0;
        }
        }
        if (!act) return;                                  // (no barrier beyond this point in the 16-bit epilogue)
    } else {
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
        ldsoff[q] = (PADB || tid + q * NT < BVEC) ? (tid + q * NT) * 16 : -1;
    }
    const unsigned wtap = (unsigned)KC16 * FRAGB;                       // bytes between consecutive taps of one channel block
    auto load_B = [&](unsigned woff, u32x4 (&r)[TPS * Q]) {             // woff = byte offset of the step's first tap (scalar)
#pragma unroll
        for (int tp = 0; tp < TPS; ++tp) {
            const size_t off = (size_t)woff + (size_t)tp * wtap;
#pragma unroll
            for (int q = 0; q < Q; ++q) r[tp * Q + q] = *(const u32x4*)(wthr[q] + off);
        }
    };
    auto store_B = [&](int bufoff, const u32x4 (&r)[TPS * Q]) {
#pragma unroll
        for (int tp = 0; tp < TPS; ++tp)
#pragma unroll
            for (int q = 0; q < Q; ++q)
                if (PADB || ldsoff[q] >= 0) *(u32x4*)(ldsB + bufoff + tp * BPITCH + ldsoff[q]) = r[tp * Q + q];
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
                for (int mb = 0; mb < MB; ++mb) {
                    if constexpr (TR) mma_k16(fb[SET][j][i], fa[SET][j][mb], acc[mb][i], T());
                    else mma_k16(fa[SET][j][mb], fb[SET][j][i], acc[mb][i], T());
                }
            }
    };
    // LDS byte shift of the NEXT tap to prefetch, advanced incrementally (kw, kh, kd counters): computing it from the
    // tap index costs two scalar divisions per tap, and the scalar unit is shared by the CU's eight wavefronts -- the
    // step loop carried ~150 scalar instructions per 24 MFMAs per wave (ISA count), enough to keep it ~80 % busy.
    // Taps run kd-major; the zero tap that pads an odd tap count and the first tap of the next slab both read shift 0.
    int nshift = 0, tidx = 0, kw_c = 0, kh_c = 0;
    auto advance_tap = [&]() {
        ++tidx;
        if (tidx >= NTAPS) {
            nshift = 0; kw_c = 0; kh_c = 0;
            if (tidx == ((TPS == 1) ? NTAPS : NTP)) tidx = 0;        // one tap per step walks the 27 real taps only
            return;
        }
        ++kw_c; nshift += PITCH;
        if (kw_c == KW) {
            kw_c = 0; nshift += (HW_ - KW) * PITCH;
            if (++kh_c == KH) { kh_c = 0; nshift += (HH_ - KH) * HW_ * PITCH; }
        }
    };

    // (slab, step-in-slab) cursors: c0 = current step, c3 = step + 4 (weight loads)
    int slab0 = 0, sis0 = 0, slab3 = 0, sis3 = 0;
    unsigned woff3 = 0;                                     // running weight offset of the step the cursor points at (no multiplies per step)
    auto adv = [&](int& sl, int& si) { if (++si == SPS) { si = 0; ++sl; } };
    auto adv_clamped = [&]() {                             // past the end: stay on (re-read) the last tile
        if (slab3 == nslab - 1 && sis3 == SPS - 1) return;
        woff3 += (unsigned)TPS * wtap;
        if (++sis3 == SPS) { sis3 = 0; ++slab3; woff3 = (unsigned)slab3 * KS * FRAGB; }
    };

    ss_load();
    stage_A(0);
    ss_store();
    load_B(0u, R0);
    adv_clamped();
    load_B(woff3, R1);
    adv_clamped();
    store_B(0, R0);
    load_B(woff3, R0);                               // step 2
    adv_clamped();
    store_B(BSTEP, R1);
    load_B(woff3, R1);                               // step 3
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
            advance_tap();
            read_frags(std::integral_constant<int, SET ^ 1>(), b0 + (TP + 1) * BPITCH, nshift);
        } else {                                           // next slot opens step s+1
            new_slab = (sis0 + 1 == SPS);
            advance_tap();                                 // (a new slab restarts at tap 0: shift 0)
            if (s_ + 1 < S && !new_slab)
                read_frags(std::integral_constant<int, SET ^ 1>(), b1, nshift);
        }
        mma_all(setc);
        if (LAST) {
            store_B(b2, R);                                // (past the end: a duplicate tile into a buffer nobody reads)
            load_B(woff3, R);
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

    }   // classic pipeline

    epilogue([]() {});
}


template <typename T, int TWL, int NB, int KD, int KH, int KW, int TPS, int MB, int WV, int PH = 0>
__global__ __launch_bounds__(WV * 64, (WV == 8 && NB == 1 && TWL == 0 && PH == 0) ? 4 : 2)
void conv_tap_kernel(ConvParams p) {
    conv_tap_body<T, TWL, NB, KD, KH, KW, TPS, MB, WV, PH>(p);
}

// The same workgroups for up to CONV_GROUP_MAX independent problems in one grid (an Inception block's branch_1 and branch_2 3x3x3
// convs: neither fills the chip's second round alone, and a launch boundary between them idles every CU for a prologue + an epilogue).
// (Measured and not taken, round 6: capping the NB = 1 instantiation at 128 VGPRs -- __launch_bounds__(512, 4), ~10 registers spilled in the
// prologue / epilogue only -- so that TWO of these workgroups fit a CU (68-74 KiB of LDS each).  Whole C2 step, variants interleaved,
// gpurun_out/ab_c3.txt: 1.2428 against 1.2382 ms one batch at a time, 1.0903 against 1.0872 ms with two in flight -- 0.3-0.4 % SLOWER.)
template <typename T, int TWL, int NB, int KD, int KH, int KW, int TPS, int MB, int WV, int PH>
__global__ __launch_bounds__(WV * 64, 2)
void conv_tap_group_kernel(ConvGroupParams g) {
    const int k = (g.n > 1 && blockIdx.x >= (unsigned)g.p[1].gbase) ? 1 : 0;
    conv_tap_body<T, TWL, NB, KD, KH, KW, TPS, MB, WV, PH, true>(g.p[k]);
}


// conv_tap_kernel with the fused pointwise input (PRE, see conv_tap_body)
template <typename T, int TWL, int NB>
__global__ __launch_bounds__(512, 2)
void conv_tap_pre_kernel(ConvParams p) {
    conv_tap_body<T, TWL, NB, 3, 3, 3, 2, 2, 8, 1, false, true>(p);
}


// ... and with maxPool3d_3a taken on the tile (POOL, see the epilogue of conv_tap_body): the 4-plane 8x8 tile only
template <typename T, int NB>
__global__ __launch_bounds__(512, 2)
void conv_tap_pre_pool_kernel(ConvParams p) {
    conv_tap_body<T, 3, NB, 3, 3, 3, 2, 2, 8, 1, false, true, true>(p);
}


// The grouped launch plus ONE pointwise conv (conv_pw_body<T, 1, 4>: 128 pixels x 64 channels per 256-thread workgroup; waves 4-7 of
// such a workgroup leave at once).  On the 14x14 maps the two 3x3x3 convs of an Inception block are 168-224 one-per-CU workgroups
// of 23-52 us: the block's branch_3 1x1x1 conv (9-11 us as a launch of its own) runs beside them on the idle CUs.
template <typename T, int TWL, int NB, int KD, int KH, int KW, int TPS, int MB, int WV, int PH>
__global__ __launch_bounds__(WV * 64, 2)
void conv_tap_group_pw_kernel(ConvGroupParams g) {
    if (blockIdx.x >= (unsigned)g.pw.gbase) {
        if (threadIdx.x < 256) conv_pw_body<T, 1, 4>(g.pw);
        return;
    }
    const int k = (g.n > 1 && blockIdx.x >= (unsigned)g.p[1].gbase) ? 1 : 0;
    conv_tap_body<T, TWL, NB, KD, KH, KW, TPS, MB, WV, PH, true>(g.p[k]);
}


template <typename T, int TWL, int KD, int KH, int KW>
static int launch_tap(const ConvParams& p, int NB, int tps, int wv, dim3 grid, step_stream_t stream) {
#define STEP_TAP(NB_, TPS_, MB_) STEP_LAUNCH((conv_tap_kernel<T, TWL, NB_, KD, KH, KW, TPS_, MB_, 8>), grid, dim3(512), stream, p)
#define STEP_TAP4(NB_, TPS_) STEP_LAUNCH((conv_tap_kernel<T, TWL, NB_, KD, KH, KW, TPS_, 2, 4>), grid, dim3(256), stream, p)
    if (wv == 4) {                     // two resident workgroups per CU: one tap per barrier (NB = 1: two, its weight tiles are small)
        switch (NB) {
            case 1: STEP_TAP4(1, 2); break;
            case 2: STEP_TAP4(2, 1); break;
            default: STEP_TAP4(3, 1); break;
        }
    } else if (tps == 2) {
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
#undef STEP_TAP4
    return STEP_LAUNCH_CHECK();
}


// two-phase form (PH = 1): 16-bit storage, 3x3x3 windows; instantiated in conv_tap_ph_{bf16,f16}.hip
// (1x3x3 windows have 5 steps per slab -- an odd count doubles the unrolled body and spills; PH = 2, the wave grouping
// that pairs waves of DIFFERENT SIMDs, measured 20-30 % slower than the classic form and is not instantiated)
template <typename T, int TWL>
static int launch_tap_ph(const ConvParams& p, int NB, dim3 grid, step_stream_t stream) {
#define STEP_TAPPH(NB_) STEP_LAUNCH((conv_tap_kernel<T, TWL, NB_, 3, 3, 3, 2, 2, 8, 1>), grid, dim3(512), stream, p)
    switch (NB) {
        case 1: STEP_TAPPH(1); break;
        case 2: STEP_TAPPH(2); break;
        default: STEP_TAPPH(3); break;
    }
#undef STEP_TAPPH
    return STEP_LAUNCH_CHECK();
}

template <typename T, int TWL>
static int launch_tap_pre(const ConvParams& p, int NB, dim3 grid, step_stream_t stream) {
#define STEP_TAPPRE(NB_) STEP_LAUNCH((conv_tap_pre_kernel<T, TWL, NB_>), grid, dim3(512), stream, p)
    switch (NB) {
        case 1: STEP_TAPPRE(1); break;
        case 2: STEP_TAPPRE(2); break;
        default: STEP_TAPPRE(3); break;
    }
#undef STEP_TAPPRE
    return STEP_LAUNCH_CHECK();
}

// ... as a persistent tile loop (PERSIST): gridDim.x workgroups walk p.gcount virtual block ids
template <typename T, int NB>
__global__ __launch_bounds__(512, 2)
void conv_tap_pre_pool_persist_kernel(ConvParams p) {
    conv_tap_body<T, 3, NB, 3, 3, 3, 2, 2, 8, 1, false, true, true, true>(p);
}

template <typename T>
static int launch_tap_pre_pool(const ConvParams& p, int NB, dim3 grid, step_stream_t stream) {
    if (p.gpersist > 0 && NB == 3) {                       // (the full rounds of a one-channel-group layer; the NB = 1 tail launch stays one tile per workgroup)
        STEP_LAUNCH((conv_tap_pre_pool_persist_kernel<T, 3>), dim3((unsigned)p.gpersist), dim3(512), stream, p);
        return STEP_LAUNCH_CHECK();
    }
    switch (NB) {
        case 1: STEP_LAUNCH((conv_tap_pre_pool_kernel<T, 1>), grid, dim3(512), stream, p); break;
        case 2: STEP_LAUNCH((conv_tap_pre_pool_kernel<T, 2>), grid, dim3(512), stream, p); break;
        default: STEP_LAUNCH((conv_tap_pre_pool_kernel<T, 3>), grid, dim3(512), stream, p); break;
    }
    return STEP_LAUNCH_CHECK();
}

// ... and 1x3x3 windows (the heads' 2-D Bottleneck convs on plane-folded general boxes, round 6): 9 taps = four two-tap steps + one single-tap
// step per slab (the padding tap is not multiplied), the odd step count alternates the ring parity per slab, so the slab body exists twice.  NB <= 2
// only: NB = 1 allocates 164 VGPRs without a spill, NB = 2 256 with 18 spilled registers that are all touched OUTSIDE the steps (ISA checked: prologue,
// slab switch, epilogue); NB = 3 spills 200 and stays on the classic form.
template <typename T>
static int launch_tap_ph_133(const ConvParams& p, int NB, dim3 grid, step_stream_t stream) {
    switch (NB) {
        case 1: STEP_LAUNCH((conv_tap_kernel<T, 0, 1, 1, 3, 3, 2, 2, 8, 1>), grid, dim3(512), stream, p); break;
        case 2: STEP_LAUNCH((conv_tap_kernel<T, 0, 2, 1, 3, 3, 2, 2, 8, 1>), grid, dim3(512), stream, p); break;
        default: return STEP_E_UNSUPPORTED;
    }
    return STEP_LAUNCH_CHECK();
}

template <typename T>
int conv_tap_ph_launch_impl(const ConvPlan& pl, const ConvParams& p, int kd, dim3 grid, step_stream_t stream) {
    if (kd == 1 && pl.ph == 1 && pl.twl == 0 && !p.pre_w) return launch_tap_ph_133<T>(p, pl.NB, grid, stream);
    if (kd != 3 || pl.ph != 1) return STEP_E_UNSUPPORTED;
    if (p.pre_w && p.pool_row) {                           // ... with the (1,3,3) / (1,2,2) max pool on the tile: the 4 x 8 x 8 tile form only
        if (pl.twl != 3) return STEP_E_UNSUPPORTED;
        return launch_tap_pre_pool<T>(p, pl.NB, grid, stream);
    }
    if (p.pre_w) {                                         // fused pointwise input: the general-box and the 4 x 8 x 8 tile forms
        if (pl.twl == 0) return launch_tap_pre<T, 0>(p, pl.NB, grid, stream);
        if (pl.twl == 3) return launch_tap_pre<T, 3>(p, pl.NB, grid, stream);
        return STEP_E_UNSUPPORTED;
    }
    if (pl.twl == 0) return launch_tap_ph<T, 0>(p, pl.NB, grid, stream);
    if (pl.twl == 3) return launch_tap_ph<T, 3>(p, pl.NB, grid, stream);
    return pl.wide ? launch_tap_ph<T, 5>(p, pl.NB, grid, stream) : launch_tap_ph<T, 4>(p, pl.NB, grid, stream);
}

// grouped launch (two-phase form; general boxes, or the 4-plane 8x8 tiles of the heads' 7x7 maps): NB = the deepest member's depth
template <typename T, int TWL>
static int conv_tap_group_launch_twl(int NB, const ConvGroupParams& g, dim3 grid, step_stream_t stream) {
#define STEP_TAPG(NB_) STEP_LAUNCH((conv_tap_group_kernel<T, TWL, NB_, 3, 3, 3, 2, 2, 8, 1>), grid, dim3(512), stream, g)
    switch (NB) {
        case 1: STEP_TAPG(1); break;
        case 2: STEP_TAPG(2); break;
        default: STEP_TAPG(3); break;
    }
#undef STEP_TAPG
    return STEP_LAUNCH_CHECK();
}
template <typename T>
int conv_tap_group_launch_impl(int twl, int NB, const ConvGroupParams& g, dim3 grid, step_stream_t stream) {
    if (g.pw.gcount > 0) {                               // with a pointwise member (general boxes only: the planner asks for nothing else)
        if (twl != 0) return STEP_E_UNSUPPORTED;
#define STEP_TAPGP(NB_) STEP_LAUNCH((conv_tap_group_pw_kernel<T, 0, NB_, 3, 3, 3, 2, 2, 8, 1>), grid, dim3(512), stream, g)
        switch (NB) {
            case 1: STEP_TAPGP(1); break;
            case 2: STEP_TAPGP(2); break;
            default: STEP_TAPGP(3); break;
        }
#undef STEP_TAPGP
        return STEP_LAUNCH_CHECK();
    }
    if (twl == 0) return conv_tap_group_launch_twl<T, 0>(NB, g, grid, stream);
    if (twl == 3) return conv_tap_group_launch_twl<T, 3>(NB, g, grid, stream);
    return STEP_E_UNSUPPORTED;
}

template <typename T>
int conv_tap_launch(const ConvPlan& pl, const ConvParams& p, int kd, dim3 grid, step_stream_t stream) {
    if constexpr (sizeof(T) == 2) {
        if (pl.ph != 0 && pl.wv == 8 && pl.tps == 2) return conv_tap_ph_launch<T>(pl, p, kd, grid, stream);
    }
    if (kd == 3) {
        if (pl.twl == 0) return launch_tap<T, 0, 3, 3, 3>(p, pl.NB, pl.tps, pl.wv, grid, stream);
        if (pl.twl == 3) return launch_tap<T, 3, 3, 3, 3>(p, pl.NB, pl.tps, pl.wv, grid, stream);
        return pl.wide ? launch_tap<T, 5, 3, 3, 3>(p, pl.NB, pl.tps, pl.wv, grid, stream) : launch_tap<T, 4, 3, 3, 3>(p, pl.NB, pl.tps, pl.wv, grid, stream);
    }
    if (pl.twl == 0) return launch_tap<T, 0, 1, 3, 3>(p, pl.NB, pl.tps, pl.wv, grid, stream);
    if (pl.twl == 3) return launch_tap<T, 3, 1, 3, 3>(p, pl.NB, pl.tps, pl.wv, grid, stream);
    return pl.wide ? launch_tap<T, 5, 1, 3, 3>(p, pl.NB, pl.tps, pl.wv, grid, stream) : launch_tap<T, 4, 1, 3, 3>(p, pl.NB, pl.tps, pl.wv, grid, stream);
}

}  // namespace step
