// step_amd/csrc/conv_pw.hip -- pointwise (1x1x1 / Linear) convs: the streaming 8-wave GEMM and the split-K path for
// few-row / very-deep-K layers.
#include "conv_common.h"
#include "conv_pw_kernel.h"

namespace step {

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
            if (cok && rok[mb]) {                          // 16 bytes per lane for 16-bit storage, 2 x 16 bytes for fp32
                typedef typename Ld16<T>::type v16;
                v16 part[sizeof(frag_t) / 16];
#pragma unroll
                for (int q = 0; q < (int)(sizeof(frag_t) / 16); ++q) part[q] = *(const v16*)(arow[mb] + ks * 16 + q * (16 / (int)sizeof(T)));
                __builtin_memcpy(&a, part, sizeof(a));
            }
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

// 256 threads = 32 output elements x 8 K-slices: slice q sums chunks q, q + 8, ... of its element (12 loads in flight per thread
// instead of one thread walking all ~100 chunks: the first form took 30 us for 0.5 MB of partial tiles, a tenth of a head step's
// launches), the eight partial sums meet in LDS and are added in a FIXED order: deterministic, no atomics.
template <typename T>
__global__ __launch_bounds__(256) void pw_splitk_finish_kernel(ConvParams p, const float* __restrict__ ws, int ksplit, int mpad, int cpad) {
    __shared__ float part[8][32];
    const long long total = p.Mtot * p.Cout;
    const int e = threadIdx.x & 31, q = threadIdx.x >> 5;
    for (long long base = (long long)blockIdx.x * 32; base < total; base += (long long)gridDim.x * 32) {
        const long long idx = base + e;
        const bool ok = idx < total;
        const long long row = ok ? idx / p.Cout : 0;
        const int co = ok ? (int)(idx % p.Cout) : 0;
        float v = 0.f;
        if (ok)
            for (int k = q; k < ksplit; k += 8) v += ws[((size_t)k * mpad + row) * cpad + co];
        part[q][e] = v;
        __syncthreads();
        if (q == 0 && ok) {
            v = ((part[0][e] + part[1][e]) + (part[2][e] + part[3][e])) + ((part[4][e] + part[5][e]) + (part[6][e] + part[7][e]));
            v = v * (p.scale ? p.scale[co] : 1.f) + (p.shift ? p.shift[co] : 0.f);
            if (p.res) v += elem<T>::to_f32(((const T*)p.res)[(size_t)row * p.r_cstride + p.r_coff + co]);
            if (p.relu) v = fmaxf(v, 0.f);
            if (p.split > 0 && co >= p.split) ((T*)p.y2)[(size_t)row * p.y2_cstride + p.y2_coff + (co - p.split)] = elem<T>::from_f32(v);
            else ((T*)p.y)[(size_t)row * p.y_cstride + p.y_coff + co] = elem<T>::from_f32(v);
        }
        __syncthreads();
    }
}


template <typename T16>
static int splitk_forward_t(const ConvPlan& pl, const ConvParams& p, float* ws, step_stream_t stream) {
    dim3 grid((unsigned)p.nblk32, (unsigned)pl.ksplit, (unsigned)pl.mtiles);
    switch (pl.mbk) {
        case 1: STEP_LAUNCH((pw_splitk_kernel<T16, 1>), grid, dim3(256), stream, p, ws, pl.kchunk16, pl.mpad, pl.cpad); break;
        case 2: STEP_LAUNCH((pw_splitk_kernel<T16, 2>), grid, dim3(256), stream, p, ws, pl.kchunk16, pl.mpad, pl.cpad); break;
        default: STEP_LAUNCH((pw_splitk_kernel<T16, 4>), grid, dim3(256), stream, p, ws, pl.kchunk16, pl.mpad, pl.cpad); break;
    }
    const long long total = p.Mtot * p.Cout;
    STEP_LAUNCH((pw_splitk_finish_kernel<T16>), dim3(flat_grid(total, 32)), dim3(256), stream, p, (const float*)ws, pl.ksplit, pl.mpad, pl.cpad);
    return STEP_LAUNCH_CHECK();
}


// ============================================================================================
// conv_pws_kernel -- SHORT-K pointwise convs (K <= 256) as a weight-stationary stream (16-bit types).  The streaming GEMM
// above stages both operands through LDS rings with a barrier per 32-channel step; with K = 64..256 (2-8 steps) its prologue,
// first HBM round trip and epilogue are exposed once per 128/256-pixel workgroup: 2.3-3.5 TB/s of activation traffic on the
// 28x28 / 56x56 maps.  Here:
//   * a workgroup (8 waves, one per CU) loads the packed weights of its NBW output-channel blocks ONCE -- [NBW][K/16] 1 KiB
//     fragments, <= 152 KiB -- and every wave then walks 32-pixel groups on its own: no barrier after the weight load;
//   * the product is taken TRANSPOSED, D^T = W X^T: the weight fragment is the A operand (rows = output channels; the packed
//     image already has that lane layout) and the activations are the B operand, whose fragment -- lane = pixel, 8 consecutive
//     input channels -- is ONE 16-byte global load from the channels-last tensor: activations never pass through LDS;
//   * a wave holds ALL K of its pixel group in registers (S steps of 64 channels: <= 64 VGPRs) and the next group's loads
//     (another S steps) are in flight while it walks the channel blocks NB at a time (16 NB accumulator registers): up to
//     16 KiB in flight per wave, every byte of the input loaded exactly once;
//   * the accumulator holds, per lane, ONE pixel and groups of 4 consecutive output channels: the epilogue (scale, shift, ReLU,
//     two destinations) pairs lanes l / l + 32 with v_permlane32_swap and stores 16-byte pieces (8 channels) straight from
//     registers, no transpose through LDS.
constexpr int PWS_LDSW = 152 * 1024;                        // weight image: up to 152 (block, 16-channel chunk) fragments
constexpr int PWS_WV = 8;
constexpr int PWS_MAXB = 16;                                // channel blocks per workgroup

// WV = 16 (round 6): sixteen waves per workgroup, ONE operand set (no prefetch of the next group: <= 128 VGPRs).  A wave's unit of work is a
// 32-pixel group; with 8 waves per CU the 28x28 maps of C2 are 3136 groups on 2048 waves -- every wave waits for the ones that got two (1.53 on
// average: a quarter of the launch is imbalance).  4096 waves take at most one group each, and the latency the prefetch hid is covered by the
// other waves of the SIMD.
// RES (round 6): a residual tensor added before the ReLU (the heads' Bottleneck conv3, 256 -> 1024 on 20-60 k rows: two_branch.py:60-84) -- each lane
// loads the 8 bytes of its pixel x 4-channel accumulator groups BEFORE the multiply loop of the pass (the loop covers the latency).
// depth of the weight-fragment ring (register quads in flight) per instantiation: what the register file leaves (no scratch)
constexpr int pws_pf(int WV, int NB, int S, bool RES) {
    return WV == 16 ? ((NB == 2 && S == 4) ? 2 : 4) : (RES ? ((NB == 3 && S == 4) ? 4 : 6) : 8);
}
template <typename T, int NB, int S, int WV = PWS_WV, bool RES = false>
__global__ __launch_bounds__(WV * 64) void conv_pws_kernel(ConvParams p, int NBW) {
    constexpr bool DBUF = WV == 8;
    static_assert(sizeof(T) == 2, "16-bit storage types");
    typedef typename frag<T>::type frag_t;
    __shared__ __attribute__((aligned(16))) unsigned char lds[PWS_LDSW + 2 * PWS_MAXB * 32 * 4];
    const int tid = threadIdx.x, lane = tid & 63, khalf = lane >> 5;
#ifdef STEP_EMUL
    const int wave = tid >> 6;
#else
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
    const int KC16 = p.nchunks32 * 2;                         // <= 4 S
    const int nbw0 = blockIdx.y * NBW;                        // the workgroup's channel blocks [nbw0, nbw0 + NBW)
    const int nbwv = min(NBW, p.nblk32 - nbw0);
    constexpr int KCP = 4 * S;                                // chunks per block in LDS: K padded to whole steps with ZERO weights,
    {                                                         // so that the chunk loop below has no run-time bound (no branches)
        const u32x4* src = (const u32x4*)((const unsigned char*)p.w + (size_t)nbw0 * KC16 * 1024);
        const u32x4 z = {0u, 0u, 0u, 0u};
        for (int v = tid; v < nbwv * KCP * 64; v += WV * 64) {
            const int blk = v / (KCP * 64), rem = v % (KCP * 64), kc = rem >> 6;
            ((u32x4*)lds)[v] = kc < KC16 ? src[(blk * KC16 + kc) * 64 + (rem & 63)] : z;
        }
        float* sc = (float*)(lds + PWS_LDSW);
        for (int i = tid; i < nbwv * 32; i += WV * 64) {
            const int co = nbw0 * 32 + i;
            sc[i] = (p.scale && co < p.Cout) ? p.scale[co] : 1.f;
            sc[PWS_MAXB * 32 + i] = (p.shift && co < p.Cout) ? p.shift[co] : 0.f;
        }
    }
    __syncthreads();
    const float* scl = (const float*)(lds + PWS_LDSW);
    const float* shl = scl + PWS_MAXB * 32;
    const long long ngroups = (p.Mtot + 31) >> 5;
    const long long gstride = (long long)gridDim.x * WV;
    const T* xg = (const T*)p.x;

    frag_t xa[DBUF ? 2 : 1][S * 4];
    auto load_group = [&](auto setc, long long g) {
        constexpr int SET = DBUF ? decltype(setc)::value : 0;
        const long long gm = g * 32 + (lane & 31);
        const bool ok = g < ngroups && gm < p.Mtot;
        const T* xp = xg + (size_t)(ok ? gm : 0) * p.x_cstride + p.x_coff + 8 * khalf;
#pragma unroll
        for (int j = 0; j < S * 4; ++j) {
            frag_t v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (ok && j * 16 + 8 * khalf < p.Cin) v = *(const frag_t*)(xp + j * 16);
            xa[SET][j] = v;
        }
    };
    auto group = [&](auto setc, long long g) {
        constexpr int SET = DBUF ? decltype(setc)::value : 0;
        const long long gm = g * 32 + (lane & 31);
        const bool ok = gm < p.Mtot;
        for (int b0 = 0; b0 < nbwv; b0 += NB) {
            f32x16 acc[NB];
#pragma unroll
            for (int i = 0; i < NB; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
            const unsigned char* wb[NB];
#pragma unroll
            for (int i = 0; i < NB; ++i) wb[i] = lds + ((size_t)min(b0 + i, nbwv - 1) * KCP * 64 + lane) * 16;   // (a surplus block repeats the last one and is not stored)
            // (16 bytes per lane in the layout of the STORES below -- 8 channels at 16 h + 8 khalf -- and v_permlane32_swap, its own inverse,
            // takes them back to the accumulators' layout in the epilogue: 8-byte loads of the 4-channel accumulator groups touched 32
            // lines for 512 bytes per instruction, and the vector L1's line rate, not HBM, bounds this kernel: tools/ubench/l1_pattern.hip)
            u32x4 rv[RES ? NB : 1][2];
            if constexpr (RES) {
                const T* rp = (const T*)p.res + (size_t)(ok ? gm : 0) * p.r_cstride + p.r_coff + nbw0 * 32 + 8 * khalf;
#pragma unroll
                for (int i = 0; i < NB; ++i)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int c8 = (b0 + i) * 32 + 16 * h;
                        u32x4 r = {0u, 0u, 0u, 0u};
                        if (ok && b0 + i < nbwv && nbw0 * 32 + c8 + 8 * khalf < p.Cout) r = *(const u32x4*)(rp + c8);
                        rv[i][h] = r;
                    }
            }
            // the weight fragments come from LDS PF multiplies ahead of their use (a ring of PF register quads, everything unrolled): written as
            // `mma(read(j, i), ...)` the compiler kept two reads in flight and every MFMA waited out the LDS latency of its own operand -- with
            // two waves per SIMD the multiply phase ran at a third of the pipe's rate (round 6: the 256 -> 1024 layer of 60 k rows without its
            // stores 47.6 us against 17 us of MFMA time; tools/ab_bench.py, profiles/r06_ab_pws_heads.txt)
            constexpr int NF = KCP * NB, PF = NF < pws_pf(WV, NB, S, RES) ? NF : pws_pf(WV, NB, S, RES);
            frag_t wf[PF];
#pragma unroll
            for (int t = 0; t < PF; ++t) wf[t] = lds_read_bfrag<T>(wb[t % NB] + (t / NB) * 1024);
#pragma unroll
            for (int t = 0; t < NF; ++t) {
                const frag_t cur = wf[t % PF];
                if (t + PF < NF) wf[t % PF] = lds_read_bfrag<T>(wb[(t + PF) % NB] + ((t + PF) / NB) * 1024);
#ifdef STEP_EXP_PWS_NOMMA                                      // (timing experiments, tools/ab_bench.py lib=...: never the product library)
                if (t < NB) acc[t][0] += (float)xa[SET][KCP - 1][0] + (float)xa[SET][0][1] + (float)cur[0];
#else
                mma_k16(cur, xa[SET][t / NB], acc[t % NB], T());
#endif
#ifndef STEP_EMUL
                __builtin_amdgcn_sched_barrier(0);            // (the scheduler otherwise sinks every read back to its use)
#endif
            }
            // epilogue (the planner sends only layers whose channel counts / offsets / pitches are multiples of 8 here): one
            // v_permlane32_swap per packed dword pair (register quads 2h, 2h + 1) hands the lower lane channels 16h .. 16h+7 and the
            // upper lane 16h+8 .. 16h+15 of its pixel (conv_tap_kernel.h, epilogue): 16-byte stores straight from registers
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                unsigned d[4][2];
                unsigned rq[4][2] = {{0u, 0u}, {0u, 0u}, {0u, 0u}, {0u, 0u}};
                if constexpr (RES) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        unsigned r0 = rv[i][h][0], r1 = rv[i][h][1], r2 = rv[i][h][2], r3 = rv[i][h][3];
                        lane32_swap(r0, r2);
                        lane32_swap(r1, r3);
                        rq[2 * h][0] = r0; rq[2 * h][1] = r1; rq[2 * h + 1][0] = r2; rq[2 * h + 1][1] = r3;
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c4 = (b0 + i) * 32 + 8 * q + 4 * khalf;
                    const f32x4 s4 = *(const f32x4*)(scl + c4), h4 = *(const f32x4*)(shl + c4);
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = acc[i][4 * q + e] * s4[e] + h4[e];
                        if constexpr (RES) v[e] += elem<T>::from_bits16((unsigned short)(rq[q][e >> 1] >> (16 * (e & 1))));
                        if (p.relu) v[e] = fmaxf(v[e], 0.f);
                    }
                    d[q][0] = (unsigned)elem<T>::bits16(v[0]) | ((unsigned)elem<T>::bits16(v[1]) << 16);
                    d[q][1] = (unsigned)elem<T>::bits16(v[2]) | ((unsigned)elem<T>::bits16(v[3]) << 16);
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    lane32_swap(d[2 * h][0], d[2 * h + 1][0]);
                    lane32_swap(d[2 * h][1], d[2 * h + 1][1]);
                    const int co = nbw0 * 32 + (b0 + i) * 32 + 16 * h + 8 * khalf;
#ifdef STEP_EXP_PWS_NOSTORE
                    if (ok && b0 + i < nbwv && co < p.Cout && d[2 * h][0] == 0x7fc17fc1u) {
#else
                    if (ok && b0 + i < nbwv && co < p.Cout) {
#endif
                        const u32x4 o = {d[2 * h][0], d[2 * h][1], d[2 * h + 1][0], d[2 * h + 1][1]};
                        if (p.split > 0 && co >= p.split) *(u32x4*)((T*)p.y2 + (size_t)gm * p.y2_cstride + p.y2_coff + (co - p.split)) = o;
                        else *(u32x4*)((T*)p.y + (size_t)gm * p.y_cstride + p.y_coff + co) = o;
                    }
                }
            }
        }
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    long long g = (long long)blockIdx.x * WV + wave;
    if (g >= ngroups) return;
    if constexpr (!DBUF) {
        for (; g < ngroups; g += gstride) {
            load_group(I0(), g);
            group(I0(), g);
        }
        return;
    }
    load_group(I0(), g);
    while (true) {
        load_group(I1(), g + gstride);                        // (past the last group: no loads)
        group(I0(), g);
        g += gstride;
        if (g >= ngroups) break;
        load_group(I0(), g + gstride);
        group(I1(), g);
        g += gstride;
        if (g >= ngroups) break;
    }
}

template <typename T, int NB>
static void conv_pws_launch_s16(int S, const ConvParams& p, int nbw, dim3 grid, step_stream_t stream) {
    switch (S) {
        case 1: STEP_LAUNCH((conv_pws_kernel<T, NB, 1, 16>), grid, dim3(1024), stream, p, nbw); break;
        case 2: STEP_LAUNCH((conv_pws_kernel<T, NB, 2, 16>), grid, dim3(1024), stream, p, nbw); break;
        case 3: STEP_LAUNCH((conv_pws_kernel<T, NB, 3, 16>), grid, dim3(1024), stream, p, nbw); break;
        default: STEP_LAUNCH((conv_pws_kernel<T, NB, 4, 16>), grid, dim3(1024), stream, p, nbw); break;
    }
}
template <typename T, int NB>
static void conv_pws_launch_res(int S, const ConvParams& p, int nbw, dim3 grid, step_stream_t stream) {
    switch (S) {
        case 1: STEP_LAUNCH((conv_pws_kernel<T, NB, 1, PWS_WV, true>), grid, dim3(PWS_WV * 64), stream, p, nbw); break;
        case 2: STEP_LAUNCH((conv_pws_kernel<T, NB, 2, PWS_WV, true>), grid, dim3(PWS_WV * 64), stream, p, nbw); break;
        case 3: STEP_LAUNCH((conv_pws_kernel<T, NB, 3, PWS_WV, true>), grid, dim3(PWS_WV * 64), stream, p, nbw); break;
        default: STEP_LAUNCH((conv_pws_kernel<T, NB, 4, PWS_WV, true>), grid, dim3(PWS_WV * 64), stream, p, nbw); break;
    }
}
template <typename T, int NB>
static void conv_pws_launch_s(int S, const ConvParams& p, int nbw, dim3 grid, step_stream_t stream) {
    switch (S) {
        case 1: STEP_LAUNCH((conv_pws_kernel<T, NB, 1>), grid, dim3(PWS_WV * 64), stream, p, nbw); break;
        case 2: STEP_LAUNCH((conv_pws_kernel<T, NB, 2>), grid, dim3(PWS_WV * 64), stream, p, nbw); break;
        case 3: STEP_LAUNCH((conv_pws_kernel<T, NB, 3>), grid, dim3(PWS_WV * 64), stream, p, nbw); break;
        default: STEP_LAUNCH((conv_pws_kernel<T, NB, 4>), grid, dim3(PWS_WV * 64), stream, p, nbw); break;
    }
}
// nbw = channel blocks per workgroup (nbw * K/16 <= 152, K <= 256); blocks per pass and K steps: pws_shape
template <typename T>
int conv_pws_launch(int nbw, const ConvParams& p, dim3 grid, step_stream_t stream) {
    int NB, S;
    pws_shape(nbw, p.nchunks32 * 2, NB, S);
    if (p.res) {                                                             // (eight waves: sixteen measured 5-40 % slower with the residual groups, r06_ab_pws_heads.txt)
        if (NB == 1) conv_pws_launch_res<T, 1>(S, p, nbw, grid, stream);
        else if (NB == 2) conv_pws_launch_res<T, 2>(S, p, nbw, grid, stream);
        else conv_pws_launch_res<T, 3>(S, p, nbw, grid, stream);
        return STEP_LAUNCH_CHECK();
    }
    // sixteen waves where eight leave the waves with 1 .. 3 groups each (a fractional count is imbalance): option conv_pws_waves 0 auto | 8 | 16
    if (pws_sixteen(p.Mtot, grid.x)) {
        if (NB == 1) conv_pws_launch_s16<T, 1>(S, p, nbw, grid, stream);
        else conv_pws_launch_s16<T, 2>(S, p, nbw, grid, stream);           // (NB <= 2: 128 VGPRs hold one operand set + two accumulator tiles)
        return STEP_LAUNCH_CHECK();
    }
    if (NB == 1) conv_pws_launch_s<T, 1>(S, p, nbw, grid, stream);
    else if (NB == 2) conv_pws_launch_s<T, 2>(S, p, nbw, grid, stream);
    else conv_pws_launch_s<T, 3>(S, p, nbw, grid, stream);
    return STEP_LAUNCH_CHECK();
}
template <>
int conv_pws_launch<float>(int, const ConvParams&, dim3, step_stream_t) { return STEP_E_UNSUPPORTED; }
template int conv_pws_launch<bf16_t>(int, const ConvParams&, dim3, step_stream_t);
template int conv_pws_launch<f16_t>(int, const ConvParams&, dim3, step_stream_t);

template <typename T>
int conv_pw_launch(int NB, int wv, const ConvParams& p, dim3 grid, step_stream_t stream) {
    if (p.x2) {                                                // two sources (step_conv_forward_cat): 16-bit storage
        if constexpr (sizeof(T) == 2) {
            if (wv == 4) {
                switch (NB) {
                    case 1: STEP_LAUNCH((conv_pw2_kernel<T, 1, 4>), grid, dim3(256), stream, p); break;
                    case 2: STEP_LAUNCH((conv_pw2_kernel<T, 2, 4>), grid, dim3(256), stream, p); break;
                    default: STEP_LAUNCH((conv_pw2_kernel<T, 3, 4>), grid, dim3(256), stream, p); break;
                }
                return STEP_LAUNCH_CHECK();
            }
            switch (NB) {
                case 1: STEP_LAUNCH((conv_pw2_kernel<T, 1, 8>), grid, dim3(512), stream, p); break;
                case 2: STEP_LAUNCH((conv_pw2_kernel<T, 2, 8>), grid, dim3(512), stream, p); break;
                default: return STEP_E_UNSUPPORTED;                                      // (conv_forward_t never plans it)
            }
            return STEP_LAUNCH_CHECK();
        } else {
            return STEP_E_UNSUPPORTED;
        }
    }
    if (wv == 4) {
        switch (NB) {
            case 1: STEP_LAUNCH((conv_pw_kernel<T, 1, 4>), grid, dim3(256), stream, p); break;
            case 2: STEP_LAUNCH((conv_pw_kernel<T, 2, 4>), grid, dim3(256), stream, p); break;
            default: STEP_LAUNCH((conv_pw_kernel<T, 3, 4>), grid, dim3(256), stream, p); break;
        }
        return STEP_LAUNCH_CHECK();
    }
    switch (NB) {
        case 1: STEP_LAUNCH((conv_pw_kernel<T, 1, 8>), grid, dim3(512), stream, p); break;
        case 2: STEP_LAUNCH((conv_pw_kernel<T, 2, 8>), grid, dim3(512), stream, p); break;
        default: STEP_LAUNCH((conv_pw_kernel<T, 3, 8>), grid, dim3(512), stream, p); break;
    }
    return STEP_LAUNCH_CHECK();
}
template <typename T>
int conv_splitk_launch(const ConvPlan& pl, const ConvParams& p, float* ws, step_stream_t stream) { return splitk_forward_t<T>(pl, p, ws, stream); }

template int conv_pw_launch<float>(int, int, const ConvParams&, dim3, step_stream_t);
template int conv_pw_launch<bf16_t>(int, int, const ConvParams&, dim3, step_stream_t);
template int conv_pw_launch<f16_t>(int, int, const ConvParams&, dim3, step_stream_t);
template int conv_splitk_launch<float>(const ConvPlan&, const ConvParams&, float*, step_stream_t);
template int conv_splitk_launch<bf16_t>(const ConvPlan&, const ConvParams&, float*, step_stream_t);
template int conv_splitk_launch<f16_t>(const ConvPlan&, const ConvParams&, float*, step_stream_t);

}  // namespace step
