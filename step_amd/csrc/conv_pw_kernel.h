// step_amd/csrc/conv_pw_kernel.h -- the streaming pointwise conv (conv_pw_kernel) as a device-function body + its kernel wrapper.
#pragma once
#include "conv_common.h"

namespace step {

// ============================================================================================
// conv_pw_kernel -- pointwise (1x1x1) convs / Linear layers with a deep K: a streaming GEMM.
// 512 threads = 8 wavefronts (4 x 2) own 256 consecutive pixels x (64*NB) channels; each wave a
// 64-pixel x (32*NB)-channel block.  One pipeline step = one 64-byte slab of input channels (32 x
// 16-bit / 16 x fp32).  BOTH operands stream: the A slab (256 pixels x 64 B, 80-byte pitch) and the
// weight tile go global -> one register set each -> three-buffer LDS rings; fragments are double-
// buffered in registers, so the ds_reads of step s+1 are issued before the MFMAs of step s and there
// is one barrier per step (the conv_tap_kernel pipeline without a resident halo tile).  Loads are
// branch-free (clamped addresses + bit masks) so the compiler keeps exact vmcnt waits in the loop.
// WV = wavefronts per workgroup: 8 (256 pixels, one resident workgroup per CU: 132 KiB of LDS at NB = 3) or 4 (128 pixels,
// <= 68 KiB: two resident workgroups per CU -- the short K loops of the Inception 1x1x1 convs (6-16 steps) are
// latency-bound with one: every workgroup's prologue, its first HBM round trip and its epilogue are exposed).
// The body is a device function so that other grids can carry pointwise workgroups (conv_tap_group_pw_kernel: a branch's 1x1x1 conv
// on the CUs a grouped 3x3x3 launch leaves idle); blocks are addressed through p.gbase / p.gcount (grid_coords), never blockIdx alone.
// EXT: the LDS comes from the caller (`arena`, conv_pw_lds_bytes<T, NB, WV>() bytes, 16-byte aligned) instead of a static array of this
// function -- a kernel that carries workgroups of several bodies declares ONE arena of the largest size (static arrays of all the
// bodies a kernel can reach are allocated side by side and would cap the occupancy of every workgroup by their SUM).
// DRX > 0 overrides the depth of the global -> register ring (see DR below).
template <typename T, int NB, int WV>
constexpr int conv_pw_lds_bytes() {
    constexpr int NT = WV * 64, ATILE = (WV / 2) * 64 * (sizeof(T) == 2 ? 64 : 80), BVEC = 2 * NB * (64 / (int)sizeof(T) / 16) * 512 * (int)sizeof(T) / 16;
    constexpr int NBUF = WV == 8 ? 4 : 3;                    // eight waves: a ring of FOUR slab buffers, one barrier per TWO K steps (see PAIR)
    return NBUF * ATILE + NBUF * ((BVEC + NT - 1) / NT) * NT * 16;
}
// TWO (round 6, conv_pw2_kernel): the input channels are the concat of two tensors -- K steps [0, p.s_split) come from p.x, the rest from
// p.x2 (the resample Bottleneck of the heads, two_branch.py:86-111: conv1 / conv2 over cat(global feature, downsampled feature) used to run
// as two accumulating launches with the 1024-channel partial sum written, rounded and re-read in between).
template <typename T, int NB, int WV, bool EXT = false, int DRX = 0, bool TWO = false>
__device__ __forceinline__ void conv_pw_body(const ConvParams& p, unsigned char* arena = nullptr) {
    static_assert(WV == 8 || WV == 4, "8 or 4 waves");
    constexpr int NT = WV * 64, WM = WV / 2, TPX = WM * 64, EROWS = WM * 32;
    constexpr int ES = (int)sizeof(T);
    constexpr int VEC = 16 / ES;
    constexpr int CKT = 64 / ES, KS = CKT / 16;
    // 16-bit storage (round 6): the slab rows sit at their natural 64-byte pitch with the four 16-byte slots of pixel p XOR-ed by (p >> 2) & 3 --
    // conflict-free for the fragment reads (16 lanes = 16 rows, one slot) AND for the staging writes (16 lanes = 4 pixels x 4 slots); the
    // padded 80-byte pitch was conflict-free for the reads only (profiles/r06_pmc_conv_pw.txt: 23 % of the LDS-active cycles were bank
    // conflicts) and 25 % larger.  fp32 keeps the padded pitch (its fragment is two slots wide).
    constexpr bool SWZ = ES == 2;
    constexpr int PITCH = SWZ ? 64 : 80;
    constexpr int ATILE = TPX * PITCH;                       // 16384 B (WV = 8) / 8192 B; fp32 20480 / 10240 B
    constexpr int FRAGB = 512 * ES, FRAGV = FRAGB / 16;
    constexpr int NBT = 2 * NB;
    constexpr int BTILE = NBT * KS * FRAGB;
    constexpr int BVEC = BTILE / 16;
    constexpr int Q = (BVEC + NT - 1) / NT;
    typedef typename frag<T>::type frag_t;

    constexpr int BSTRIDE = Q * NT * 16;                    // weight buffer pitch: every thread stores all its Q vectors (no predicate)
    // PAIR (round 6, the eight-wave form): the slab ring in LDS is FOUR deep and the loop synchronises once per TWO K steps -- a step is only
    // 8 NB / 2 MFMAs per wave, and with one barrier each the heads' deep pointwise GEMMs (K = 832 .. 1088 on 20-60 k rows) ran at 610-660 TFLOP/s
    // against the ~1 PFLOP/s of the 3x3x3 kernel (24 MFMAs between two phase switches).  Same slabs, same order of the MFMAs: bit-identical.
    constexpr bool PAIR = WV == 8;
    constexpr int NBUF = PAIR ? 4 : 3;
    static_assert(!PAIR || !EXT, "the four-deep ring needs the kernel's own LDS");
    static_assert(conv_pw_lds_bytes<T, NB, WV>() == NBUF * ATILE + NBUF * BSTRIDE, "conv_pw_lds_bytes");
    unsigned char* lds;
    if constexpr (EXT) {
        lds = arena;
    } else {
        __shared__ __attribute__((aligned(16))) unsigned char own[NBUF * ATILE + NBUF * BSTRIDE];
        lds = own;
    }
    unsigned char* const ldsA = lds;
    unsigned char* const ldsB = lds + NBUF * ATILE;

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
    const long long m0 = (long long)gbx * TPX;
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
        const int v = tid + q * NT;
        const int pix = v >> 2, slot = v & 3;
        const long long gm = m0 + pix;
        const bool ok = gm < p.Mtot;
        athr[q] = xg + ((size_t)(ok ? gm : 0) * p.x_cstride + p.x_coff) * ES;
        amask[q] = ok ? 0xffffffffu : 0u;
        acol[q] = slot * VEC;
    }
    // TWO: the second source's row, moved back by the first source's K extent so that one channel offset serves both
    const unsigned char* athr2[TWO ? 2 : 1];
    if constexpr (TWO) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const long long gm = m0 + ((tid + q * NT) >> 2);
            athr2[q] = (const unsigned char*)p.x2 + ((size_t)(gm < p.Mtot ? gm : 0) * p.x2_cstride + p.x2_coff) * ES - (size_t)p.s_split * CKT * ES;
        }
    }
    // B: this thread's vectors of a step tile (as conv_tap_kernel)
    const unsigned char* wthr[Q];
    int ldsoff[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int v = min(tid + q * NT, BVEC - 1);
        const int f = v / FRAGV, within = v % FRAGV;
        const int nbl = f / KS, ks = f % KS;
        const int nbg = min(nb0 + nbl, p.nblk32 - 1);
        wthr[q] = wg + ((size_t)nbg * KC16 + ks) * FRAGB + within * 16;
        ldsoff[q] = (tid + q * NT) * 16;
    }
    // global -> register ring of DR step slabs -> LDS ring of 3: a slab is loaded DR steps before it is written to
    // LDS (4 steps of matrix work cover the HBM latency; with one register set the load -> store distance was a
    // single step and the K loop ran latency-bound)
    // (four-wave NB = 3: 3 weight vectors per thread per step -- four register sets would spill; two suffice when a second
    // resident workgroup covers the latency)
    constexpr int DR = DRX > 0 ? DRX : ((WV == 4 && NB == 3) ? 2 : 4);
    static_assert(!PAIR || DR == 4, "the paired loop keeps slab k in register set k % 4 and LDS buffer k % 4");
    u32x4 RA[DR][2], RB[DR][Q];
    // FULL = whole slabs (Cin % CKT == 0) and a whole 256-pixel tile: no channel / pixel masks anywhere in the loop
    // (workgroup-uniform; the vector ALU work per step drops by two thirds)
    const bool full_tile = (p.Cin % CKT) == 0 && m0 + TPX <= p.Mtot;
    auto load_step = [&](auto rc, int s_, auto fullc) {
        constexpr int RS = decltype(rc)::value;
        constexpr bool FULL = decltype(fullc)::value;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (FULL) {
                const int sc = min(s_, S - 1);                                                          // past the end: re-read the last slab
                const unsigned char* src = athr[q];
                if constexpr (TWO) src = sc >= p.s_split ? athr2[q] : src;
                RA[RS][q] = *(const u32x4*)(src + (size_t)(sc * CKT + acol[q]) * ES);
            } else {
                const int c = s_ * CKT + acol[q];
                const bool cok = c < p.Cin;                               // whole vector in or out (Cin % VEC == 0)
                const unsigned char* src = athr[q];
                if constexpr (TWO) src = (cok && s_ >= p.s_split) ? athr2[q] : src;
                RA[RS][q] = *(const u32x4*)(src + (size_t)(cok ? c : 0) * ES);     // masked when it is written to LDS: an
            }                                                                       // AND here would wait for the load at once
        }
        const size_t off = (size_t)(min(s_, S - 1) * KS) * FRAGB;       // past the end: a harmless re-read of the last tile
#pragma unroll
        for (int q = 0; q < Q; ++q) RB[RS][q] = *(const u32x4*)(wthr[q] + off);
    };
    auto store_step = [&](auto rc, int buf, int slab, auto fullc) {
        constexpr int RS = decltype(rc)::value;
        constexpr bool FULL = decltype(fullc)::value;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int v = tid + q * NT;
            if (FULL) {
                *(u32x4*)(ldsA + buf * ATILE + (v >> 2) * PITCH + (((v & 3) ^ (SWZ ? ((v >> 4) & 3) : 0)) << 4)) = RA[RS][q];
            } else {
                const unsigned int mk = (slab * CKT + acol[q] < p.Cin) ? amask[q] : 0u;
                *(u32x4*)(ldsA + buf * ATILE + (v >> 2) * PITCH + (((v & 3) ^ (SWZ ? ((v >> 4) & 3) : 0)) << 4)) = RA[RS][q] & mk;
            }
        }
#pragma unroll
        for (int q = 0; q < Q; ++q)
            *(u32x4*)(ldsB + buf * BSTRIDE + ldsoff[q]) = RB[RS][q];
    };

    const unsigned char* abase[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) abase[mb] = ldsA + (wm * 64 + mb * 32 + (lane & 31)) * PITCH + (SWZ ? 0 : khalf * (ES == 4 ? 32 : 16));
    int aoff[KS];                                            // byte offset of this lane's fragment of k16 step j inside its pixel's slab row
#pragma unroll
    for (int j = 0; j < KS; ++j) aoff[j] = SWZ ? (((2 * j + khalf) ^ ((lane >> 2) & 3)) << 4) : j * 32;
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
            for (int mb = 0; mb < 2; ++mb) fa[SET][j][mb] = lds_read_bfrag<T>(abase[mb] + buf * ATILE + aoff[j]);
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
    typedef std::integral_constant<int, 2 % DR> I2;
    typedef std::integral_constant<int, 3 % DR> I3;
    int b1 = 1, b2 = 2, s_ = 0;
    // step s: register set (s + 2) % DR holds slab s + 2 -> LDS buffer (s + 2) % 3, then reloads slab s + 2 + DR.
    // No predicates inside (loads past the end are clamped and masked, the surplus fragment read hits a valid
    // buffer): any branch in the loop makes the compiler fall back to vmcnt(0) waits.
    auto run = [&](auto fullc) {
        // prologue: steps 0..DR-1 in flight at once, 0 and 1 to LDS, DR and DR+1 take their register sets
        load_step(I0(), 0, fullc); load_step(I1(), 1, fullc);
        if (DR == 4) { load_step(I2(), 2, fullc); load_step(I3(), 3, fullc); }
        store_step(I0(), 0, 0, fullc);
        store_step(I1(), 1, 1, fullc);
        load_step(I0(), DR, fullc); load_step(I1(), DR + 1, fullc);
        __syncthreads();
        read_frags(I0(), 0);
        if constexpr (PAIR) {
            // two steps per barrier: slab k lives in register set k % 4 and LDS buffer k % 4.  While the waves multiply slabs s and s + 1
            // (buffers s, s + 1) they fill buffers s + 2 and s + 3 -- last read two barriers ago.
            auto pair = [&](auto ra, auto rb) {
                constexpr int A = decltype(ra)::value, B = decltype(rb)::value;          // (s + 2) % 4, (s + 3) % 4
                read_frags(I1(), (A + 3) & 3);                                           // slab s + 1
                mma_all(I0());                                                           // slab s
                store_step(ra, A, s_ + 2, fullc);
                store_step(rb, B, s_ + 3, fullc);
                load_step(ra, s_ + 2 + DR, fullc);
                load_step(rb, s_ + 3 + DR, fullc);
                __syncthreads();
                read_frags(I0(), A);                                                     // slab s + 2 (a surplus read past the end hits a valid buffer)
                mma_all(I1());                                                           // slab s + 1
                s_ += 2;
            };
#pragma unroll 1
            while (s_ + 4 <= S) {
                pair(I2(), I3());
                pair(I0(), I1());
            }
            if (s_ + 2 <= S) pair(I2(), I3());
            if (s_ < S) mma_all(I0());                                                   // an odd last slab: its fragments are in set 0
            __syncthreads();                                                             // (the epilogue re-uses the ring: every wave's last fragment read -- issued BEHIND the last barrier -- has landed)
            return;
        }
        auto step = [&](auto setc, auto rc) {
            constexpr int SET = decltype(setc)::value;
            read_frags(std::integral_constant<int, SET ^ 1>(), b1);
            mma_all(setc);
            store_step(rc, b2, s_ + 2, fullc);
            load_step(rc, s_ + 2 + DR, fullc);
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
        if (s_ < S) {                                        // 1..3 remaining steps
            step(I0(), I2());
            if (s_ < S) step(I1(), I3());
            if (s_ < S) step(I0(), I0());
        }
    };
    if (full_tile) run(std::true_type());
    else run(std::false_type());

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
            for (int idx = tid; idx < EROWS * G; idx += NT) {
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

template <typename T, int NB, int WV>
__global__ __launch_bounds__(WV * 64, 2) void conv_pw_kernel(ConvParams p) { conv_pw_body<T, NB, WV>(p); }
// (not instantiated at NB = 3 with eight waves: that form sits at 256 VGPRs without the second source's row pointers -- conv_forward_t plans NB = 2)
template <typename T, int NB, int WV>
__global__ __launch_bounds__(WV * 64, 2) void conv_pw2_kernel(ConvParams p) { conv_pw_body<T, NB, WV, false, 0, true>(p); }

}  // namespace step
