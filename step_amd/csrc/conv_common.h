// step_amd/csrc/conv_common.h -- shared by the conv translation units (conv_igemm.hip, conv_tap_*.hip, conv_pw.hip,
// conv_wgrad.hip, stem.hip): kernel parameter blocks, the XCD-aware grid remap, LDS / fragment helpers, the launch
// plan and the cross-unit launch entry points.
#pragma once
#include "options.h"
#include "common.h"
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
#include <utility>

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
    int gmode;           // general box: 0 = accumulator rows walk the box linearly, 1 = every 16-lane LDS read group is one run of 16 columns of one box row
    int gx, gy;          // logical grid: gx pixel tiles x gy channel groups (launched as a 1-D grid, see grid_coords)
    int gbase, gcount;   // this problem's workgroups are blockIdx.x in [gbase, gbase + gcount) (gcount % 8 == 0 when remapped): the whole grid, or one
                         // member's share of a grouped launch (step_conv_forward_group)
    int tile0;           // conv_tap_kernel: first pixel tile of this launch (a layer may be launched in two parts, see conv_forward_t)
    // step_conv_forward_cat (conv_pw2_kernel): a pointwise conv over the channel CONCAT of two tensors without materialising it -- K steps
    // [0, s_split) read x, the rest x2 (same pixels); x2 = null everywhere else
    const void* x2; int x2_cstride, x2_coff, s_split;
    int gpersist;        // > 0: launch the PERSISTENT form with this many workgroups (each walks virtual ids id, id + gpersist, ... < gcount); 0: one workgroup per id
    int nchunks;   // ceil(Cin / 32)
    int nchunks32; // same (the packed-weight K extent is 2*nchunks32 k16 blocks)
    int vec_epi;   // 16-byte output stores are legal (channel strides/offsets % 8 == 0, pointers 16-B aligned)
    int nblk32;    // ceil(Cout / 32)
    long long Mtot;  // N*D*H*W
    // general boxes: ceil(2^32 / d) for d = halo rows x halo columns and d = halo columns -- the halo index tables divide by these
    // run-time extents once per staged vector; (x * m) >> 32 is exact for x < 2^16, d < 2^11
    unsigned mag_hhw, mag_hw;
    // step_conv_forward_pre (conv_tap_pre_kernel): a pointwise conv + affine + ReLU applied to the halo while it is staged (64 -> 64
    // channels: conv3d_2b in front of conv3d_2c); pre_w = its packed weights, null otherwise
    const void* pre_w; const float* pre_scale; const float* pre_shift;
    // step_conv_forward_pre_pool (conv_tap_pre_pool_kernel): a (1,3,3) / (1,2,2) max pool taken on the tile; y is then the pooled tensor
    // [N, D, Hp, Wp, C]; pool_row / pool_col receive the tiles' first rows / columns for pool_seam_fix_kernel (null: no pooling)
    void* pool_row; void* pool_col; int Hp, Wp;
    int pool_p2;   // the launch was planned with general boxes off (the 4 x 8 x 8 tile although the planner alone would take a box)
#ifdef STEP_PROBE
    unsigned long long* probe;   // tools/timeline_probe.py build only: 16 timestamp / id slots per workgroup (see probe_mark)
#endif
};

// Several independent convs launched as ONE grid (step_conv_forward_group): the members share the kernel instantiation, each has its own
// parameter block and a contiguous share of blockIdx.x (p[k].gbase ascending, p[0].gbase == 0).
// pool.hip: the block's 3x3x3 / 1 max pool and a pointwise conv (conv_pw_body<T, 1, 4>, parameter block cp with gx / gy set) as one grid
int pool333_pw_launch(int dtype, const void* x, int N, int D, int H, int W, int C, int x_cstride, int x_coff, void* y, int y_cstride, int y_coff,
                      ConvParams cp, long long conv_blocks, int nbc, step_stream_t stream);

constexpr int CONV_GROUP_MAX = 2;
// pw (conv_tap_group_pw_kernel only): a pointwise conv whose 256-thread workgroups follow the members' in the same grid (pw.gbase =
// the members' total) -- a branch's 1x1x1 conv running on the CUs the 3x3x3 members leave idle
struct ConvGroupParams { ConvParams p[CONV_GROUP_MAX]; int n; ConvParams pw; };

#ifdef STEP_PROBE
// Timeline probe (make PROBE=1 -> libstep_amd_probe.so, never the product library): wave 0 of a workgroup stores the 100 MHz
// real-time counter at phase boundaries and the hardware ids of the CU it runs on.
extern unsigned long long* g_probe_buf;       // set by step_probe_set()
__device__ __forceinline__ void probe_mark(unsigned long long* probe, int slot) {
    if (probe && threadIdx.x == 0) probe[(size_t)blockIdx.x * 16 + slot] = __builtin_amdgcn_s_memrealtime();
}
// shader-clock reading (s_memtime counts shader cycles, s_memrealtime 100 MHz): slot 14 holds entry clock at first, then
// probe_clock_end() turns it into the cycles this workgroup lived -- cycles / (slot 4 - slot 0) = the clock the CU really ran at
__device__ __forceinline__ void probe_clock_begin(unsigned long long* probe) {
    if (probe && threadIdx.x == 0) probe[(size_t)blockIdx.x * 16 + 14] = __builtin_amdgcn_s_memtime();
}
__device__ __forceinline__ void probe_clock_end(unsigned long long* probe) {
    if (probe && threadIdx.x == 0) probe[(size_t)blockIdx.x * 16 + 14] = __builtin_amdgcn_s_memtime() - probe[(size_t)blockIdx.x * 16 + 14];
}
__device__ __forceinline__ void probe_ids(unsigned long long* probe) {
    probe_clock_begin(probe);
    if (probe && threadIdx.x == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);        // HW_REG_HW_ID
        const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);      // HW_REG_XCC_ID
        probe[(size_t)blockIdx.x * 16 + 15] = ((unsigned long long)xcc << 32) | hw;
    }
}
#define STEP_PROBE_MARK(p, slot) probe_mark((p).probe, slot)
#define STEP_PROBE_IDS(p) probe_ids((p).probe)
#else
#define STEP_PROBE_MARK(p, slot)
#define STEP_PROBE_IDS(p)
#endif

// Launch order -> XCD.  Workgroup ids go round-robin over the 8 XCDs (each with its own L2), so with a plain 2-D grid
// the channel groups of one pixel tile -- which read the SAME activations -- and spatially adjacent tiles -- which
// share halos -- end up on different L2s and every one of them fetches its input from HBM / MALL again (PMC on the
// 3c fused 1x1x1: FETCH 178 MB against 51 MB of input, three channel groups).  The grid is therefore 1-D, padded to
// a multiple of 8, and remapped: ids that are consecutive on one XCD walk the channel groups of a tile first, then
// the neighbouring tiles.  Returns false for the padding workgroups (they exit before any barrier).
__device__ __forceinline__ bool grid_coords_of(const ConvParams& p, unsigned id, int& bx, int& by) {      // id in [0, p.gcount)
    const unsigned G = (unsigned)p.gcount;
    const unsigned L = (G & 7) ? id : (id & 7) * (G >> 3) + (id >> 3);
    if (L >= (unsigned)p.gx * (unsigned)p.gy) return false;
    bx = (int)(L / (unsigned)p.gy);
    by = (int)(L % (unsigned)p.gy);
    return true;
}
__device__ __forceinline__ bool grid_coords(const ConvParams& p, int& bx, int& by) {
    return grid_coords_of(p, blockIdx.x - (unsigned)p.gbase, bx, by);
}

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N - 1>{})
template <int... I, typename F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

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
constexpr int CONV_GEN_NPIX = 768;         // LDS halo pixels reserved for a general (runtime-shaped) tile: 60 KiB
constexpr int CONV_GEN_NPIX_SMALL = 640;   // ... in the NB = 1 instantiation: 50 KiB + 24 KiB of weights = two workgroups
                                           // per CU (the small 14x14 / 7x7 layers are latency-bound with one)

// WV = 4 (two resident workgroups per CU): 36 KiB (NB = 3) / 24 KiB of weight step buffers leave 44 / 56 KiB of the 80 KiB
// per-workgroup budget for the halo of a general box of <= 128 pixels
constexpr int CONV_GEN_NPIX4 = 544;
constexpr int CONV_GEN_NPIX4_WIDE = 688;
__host__ __device__ constexpr int conv_gen_npix(int wv, int nb) {
    return wv == 8 ? (nb == 1 ? CONV_GEN_NPIX_SMALL : CONV_GEN_NPIX) : (nb >= 3 ? CONV_GEN_NPIX4 : CONV_GEN_NPIX4_WIDE);
}

static inline unsigned flat_grid(long long total, int block) {
    long long g = ceil_div64(total, block);
    if (g > 16384) g = 16384;
    return (unsigned)g;
}


struct ConvPlan { bool ok, flat, wide, deep; int impl, NB, tps, mb, wv, ph, twl, tiles_h, tiles_w, tiles_d, gtd, gth, gtw, gmode, ksplit, kchunk16, mbk, mpad, cpad; long long mtiles; };

static inline int conv_impl_override() {   // STEP_OPT_CONV_IMPL: -1 auto, 0 tiled, 1 conv_tap (one tap per step), 2 conv_tap, 5 streaming pointwise
    return opt(STEP_OPT_CONV_IMPL);
}

// launchers defined in the other translation units (explicitly instantiated for float, bf16_t, f16_t)
template <typename T> int conv_tap_launch(const ConvPlan& pl, const ConvParams& p, int kd, dim3 grid, step_stream_t stream);   // conv_tap_<dtype>.hip
template <typename T> int conv_tap_ph_launch(const ConvPlan& pl, const ConvParams& p, int kd, dim3 grid, step_stream_t stream);  // conv_tap_ph_<dtype>.hip (two-phase form)
template <> int conv_tap_ph_launch<bf16_t>(const ConvPlan& pl, const ConvParams& p, int kd, dim3 grid, step_stream_t stream);
template <> int conv_tap_ph_launch<f16_t>(const ConvPlan& pl, const ConvParams& p, int kd, dim3 grid, step_stream_t stream);
template <typename T> int conv_tap_group_launch(int twl, int NB, const ConvGroupParams& g, dim3 grid, step_stream_t stream);       // conv_tap_ph_<dtype>.hip
template <> int conv_tap_group_launch<bf16_t>(int twl, int NB, const ConvGroupParams& g, dim3 grid, step_stream_t stream);
template <> int conv_tap_group_launch<f16_t>(int twl, int NB, const ConvGroupParams& g, dim3 grid, step_stream_t stream);
template <typename T> int conv_pw_launch(int NB, int wv, const ConvParams& p, dim3 grid, step_stream_t stream);                        // conv_pw.hip
template <typename T> int conv_pws_launch(int nbw, const ConvParams& p, dim3 grid, step_stream_t stream);                              // conv_pw.hip (weight-stationary stream, 16-bit)
// conv_pws_kernel: nbw channel blocks per workgroup, KC16 16-channel chunks -> blocks per pass (<= 3: registers), 64-channel steps
// sixteen waves per workgroup (conv_pw.hip, conv_pws_kernel<.., 16>) where eight would leave every wave with 1 .. 3 pixel groups: `gx` = pixel-axis
// workgroups of the launch, M = rows.  STEP_OPT_CONV_PWS_WAVES: 0 auto | 8 | 16
// Measured (round 6, profiles/r06_ab_pws_waves.txt): 28x28 maps of 8 clips (3136 groups on 2048 | 4096 waves) 3b 24.6 -> 21.8 us, 3c 35.4 -> 32.5 us, the
// C2 step one batch at a time 1.2710 -> 1.2559 ms; 50x50 maps of 4 clips (2.75 groups per wave at eight: already balanced) 4-9 % SLOWER, and with two
// batches in flight the step is 1.6 % slower (the other batch fills the imbalance anyway, and 1024-thread workgroups leave it no room): so only where
// sixteen waves give every wave at most ONE group and eight do not, and not under the throughput profile.
// Measured (round 6, profiles/r06_ab_pws_waves.txt): 28x28 maps of 8 clips (3136 groups on 2048 | 4096 waves) 3b 24.6 -> 21.8 us, 3c 35.4 -> 32.5 us, the
// C2 step one batch at a time 1.2710 -> 1.2559 ms; 50x50 maps of 4 clips (2.75 groups per wave at eight: already balanced) 4-9 % SLOWER, and with two
// batches in flight the step is 1.6 % slower (the other batch fills the imbalance anyway, and 1024-thread workgroups leave it no room): so only where
// sixteen waves give every wave at most ONE group and eight do not, and not under the throughput profile.
// the residual form (round 6): sixteen waves on request only until measured
static inline bool pws_sixteen_res(long long M, long long gx) {
    (void)M; (void)gx;
    return opt(STEP_OPT_CONV_PWS_WAVES) == 16;
}
static inline bool pws_sixteen(long long M, long long gx) {
    const int o = opt(STEP_OPT_CONV_PWS_WAVES);
    const long long g = (M + 31) >> 5;
    return o == 16 || (o == 0 && opt(STEP_OPT_THROUGHPUT) == 0 && g > gx * 8 && g <= gx * 16);
}
static inline void pws_shape(int nbw, int KC16, int& NB, int& S) {
    NB = nbw >= 3 ? 3 : nbw;
    if (nbw == 4) NB = 2;                                     // two even passes
    S = (KC16 + 3) / 4;
}
template <typename T> int conv_splitk_launch(const ConvPlan& pl, const ConvParams& p, float* ws, step_stream_t stream);        // conv_pw.hip

}  // namespace step
