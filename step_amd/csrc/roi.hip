// step_amd/csrc/roi.hip -- ROIAlign / ROIPool forward + backward for gfx950.
//
// From-scratch replacements for the reference's CUDA operators
//   external/maskrcnn_benchmark/csrc/cuda/ROIAlign_cuda.cu  (RoIAlignForward :88-146,
//       RoIAlignBackwardFeature :201-278, bilinear_interpolate :39-86,149-199)
//   external/maskrcnn_benchmark/csrc/cuda/ROIPool_cuda.cu   (RoIPoolFForward :40-101,
//       RoIPoolFBackward :103-132)
//
// MI355X design.  The reference maps one thread to one output SCALAR of an NCHW tensor, so the
// four bilinear taps of every sample are scattered 4-byte reads with the channel as the slowest
// index.  Here the native layout is channels-last ([B,H,W,C]): one workgroup owns one
// (roi, bin), its lanes run along C in 16-byte vectors, and every tap is one fully coalesced
// row read (C = 832 fp32 -> 3.3 KB contiguous per tap).  The op is gather/HBM-bound; there is
// nothing for MFMA to do.  An NCHW path (thread per scalar) is kept for callers that hand over
// torch-contiguous tensors, with the same arithmetic.
//
// Arithmetic follows the reference operation for operation (no FMA contraction in this file),
// so fp32 results are bit-identical to cpu/ROIAlign_cpu.cpp.
#include "common.h"
#include "options.h"

#pragma clang fp contract(off)

namespace step {

struct RoiGeom {
    int batch;
    float start_w, start_h, bin_h, bin_w;
    int grid_h, grid_w;
    float count;
};

// ROIAlign_cuda.cu:101-124 / ROIAlign_cpu.cpp:160-189
__device__ __forceinline__ RoiGeom roi_geom(const float* r, float scale, int ph, int pw, int sampling_ratio) {
    RoiGeom g;
    g.batch = (int)r[0];
    g.start_w = r[1] * scale;  // no rounding
    g.start_h = r[2] * scale;
    float end_w = r[3] * scale;
    float end_h = r[4] * scale;
    float roi_w = fmaxf(end_w - g.start_w, 1.f);  // malformed ROIs are forced to 1x1
    float roi_h = fmaxf(end_h - g.start_h, 1.f);
    g.bin_h = roi_h / (float)ph;
    g.bin_w = roi_w / (float)pw;
    g.grid_h = (sampling_ratio > 0) ? sampling_ratio : (int)ceilf(roi_h / (float)ph);
    g.grid_w = (sampling_ratio > 0) ? sampling_ratio : (int)ceilf(roi_w / (float)pw);
    g.count = (float)(g.grid_h * g.grid_w);
    return g;
}

struct Tap {
    int y_low, y_high, x_low, x_high;  // -1 => void sample
    float w1, w2, w3, w4;
};

// ROIAlign_cuda.cu:39-86 / :149-199
__device__ __forceinline__ Tap bilinear_tap(int height, int width, float y, float x) {
    Tap t;
    if (y < -1.0f || y > (float)height || x < -1.0f || x > (float)width) {
        t.y_low = t.y_high = t.x_low = t.x_high = -1;
        t.w1 = t.w2 = t.w3 = t.w4 = 0.f;
        return t;
    }
    if (y <= 0) y = 0;
    if (x <= 0) x = 0;
    int y_low = (int)y, x_low = (int)x, y_high, x_high;
    if (y_low >= height - 1) { y_high = y_low = height - 1; y = (float)y_low; } else { y_high = y_low + 1; }
    if (x_low >= width - 1) { x_high = x_low = width - 1; x = (float)x_low; } else { x_high = x_low + 1; }
    float ly = y - (float)y_low, lx = x - (float)x_low;
    float hy = 1.f - ly, hx = 1.f - lx;
    t.w1 = hy * hx; t.w2 = hy * lx; t.w3 = ly * hx; t.w4 = ly * lx;
    t.y_low = y_low; t.y_high = y_high; t.x_low = x_low; t.x_high = x_high;
    return t;
}

__device__ __forceinline__ float sample_y(const RoiGeom& g, int p, int iy) {
    return g.start_h + (float)p * g.bin_h + ((float)iy + .5f) * g.bin_h / (float)g.grid_h;
}
__device__ __forceinline__ float sample_x(const RoiGeom& g, int q, int ix) {
    return g.start_w + (float)q * g.bin_w + ((float)ix + .5f) * g.bin_w / (float)g.grid_w;
}

template <typename T, int V> struct VecIO {
    __device__ static __forceinline__ void load(const T* p, float (&f)[V]) {
#pragma unroll
        for (int i = 0; i < V; ++i) f[i] = elem<T>::to_f32(p[i]);
    }
    __device__ static __forceinline__ void store(T* p, const float (&f)[V]) {
#pragma unroll
        for (int i = 0; i < V; ++i) p[i] = elem<T>::from_f32(f[i]);
    }
};
// 16-byte specialisations
template <> struct VecIO<float, 4> {
    __device__ static __forceinline__ void load(const float* p, float (&f)[4]) {
        f32x4 v = *(const f32x4*)p;
        f[0] = v[0]; f[1] = v[1]; f[2] = v[2]; f[3] = v[3];
    }
    __device__ static __forceinline__ void store(float* p, const float (&f)[4]) {
        f32x4 v = {f[0], f[1], f[2], f[3]};
        *(f32x4*)p = v;
    }
};
template <typename T> struct VecIO16 {
    __device__ static __forceinline__ void load(const T* p, float (&f)[8]) {
        u16x8 v = *(const u16x8*)p;
#pragma unroll
        for (int i = 0; i < 8; ++i) { T e; e.v = v[i]; f[i] = elem<T>::to_f32(e); }
    }
    __device__ static __forceinline__ void store(T* p, const float (&f)[8]) {
        u16x8 v;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = elem<T>::from_f32(f[i]).v;
        *(u16x8*)p = v;
    }
};
template <> struct VecIO<bf16_t, 8> : VecIO16<bf16_t> {};
template <> struct VecIO<f16_t, 8> : VecIO16<f16_t> {};

// ---------------------------------------------------------------------------------------------
// ROIAlign forward, channels-last.  grid = K*ph*pw workgroups, lanes along C in vectors of V.
template <typename T, int V>
__global__ void roi_align_fwd_nhwc_kernel(const T* __restrict__ feat, const float* __restrict__ rois, int C, int H,
                                          int W, int ph, int pw, float scale, int sampling_ratio,
                                          T* __restrict__ out, int tsl, int tall) {
    const int bin = blockIdx.x;
    const int n = bin / (ph * pw);
    const int p = (bin / pw) % ph;
    const int q = bin % pw;
    const RoiGeom g = roi_geom(rois + 5 * n, scale, ph, pw, sampling_ratio);
    // tube form (step_roi_align_tubes_forward): the roi's frame index counts the frames of a T-slice [t0, t0 + tsl) of every
    // clip (b * tsl + t); the feature buffer holds tall frames per clip and `feat` points at frame t0 of clip 0
    const int frame = tsl > 0 ? (g.batch / tsl) * tall + g.batch % tsl : g.batch;
    const T* base = feat + (size_t)frame * H * W * C;
    for (int c = threadIdx.x * V; c < C; c += blockDim.x * V) {
        float acc[V];
#pragma unroll
        for (int i = 0; i < V; ++i) acc[i] = 0.f;
        for (int iy = 0; iy < g.grid_h; ++iy) {
            const float y = sample_y(g, p, iy);
            for (int ix = 0; ix < g.grid_w; ++ix) {
                const float x = sample_x(g, q, ix);
                const Tap t = bilinear_tap(H, W, y, x);
                if (t.y_low < 0) continue;  // contributes exactly 0
                float v1[V], v2[V], v3[V], v4[V];
                VecIO<T, V>::load(base + ((size_t)t.y_low * W + t.x_low) * C + c, v1);
                VecIO<T, V>::load(base + ((size_t)t.y_low * W + t.x_high) * C + c, v2);
                VecIO<T, V>::load(base + ((size_t)t.y_high * W + t.x_low) * C + c, v3);
                VecIO<T, V>::load(base + ((size_t)t.y_high * W + t.x_high) * C + c, v4);
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    float val = t.w1 * v1[i] + t.w2 * v2[i] + t.w3 * v3[i] + t.w4 * v4[i];
                    acc[i] += val;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < V; ++i) acc[i] /= g.count;
        VecIO<T, V>::store(out + (size_t)bin * C + c, acc);
    }
}

// (round 6: a 16-bit variant with the loads of up to four samples of a row in flight at once measured SLOWER -- 132.7 against 90.4 us per
// launch at 4 x 34 tubes, C3 670.6 against 686.6 clips/s, profiles/r06_ab_pws_heads.txt: the adaptive grid of the tubes is mostly 2 x 2, so
// half the batched loads were padding, and 64 more VGPRs cost occupancy; the same with the bin's samples FLATTENED so that a 2 x 2 grid is
// exactly one batch of 16 loads, no padding: 674.6 / 679.6 against 685.3 / 688.4 clips/s -- the kernel is bound by the L1 / L2 throughput of
// its taps at full occupancy, not by a dependent-latency chain.  The loop above stays.)
// ROIAlign forward, NCHW (thread per output scalar; same arithmetic).
template <typename T>
__global__ void roi_align_fwd_nchw_kernel(const T* __restrict__ feat, const float* __restrict__ rois, long long total,
                                          int C, int H, int W, int ph, int pw, float scale, int sampling_ratio,
                                          T* __restrict__ out) {
    for (long long index = (long long)blockIdx.x * blockDim.x + threadIdx.x; index < total;
         index += (long long)blockDim.x * gridDim.x) {
        const int q = (int)(index % pw);
        const int p = (int)((index / pw) % ph);
        const int c = (int)((index / pw / ph) % C);
        const int n = (int)(index / pw / ph / C);
        const RoiGeom g = roi_geom(rois + 5 * n, scale, ph, pw, sampling_ratio);
        const T* d = feat + ((size_t)g.batch * C + c) * H * W;
        float acc = 0.f;
        for (int iy = 0; iy < g.grid_h; ++iy) {
            const float y = sample_y(g, p, iy);
            for (int ix = 0; ix < g.grid_w; ++ix) {
                const float x = sample_x(g, q, ix);
                const Tap t = bilinear_tap(H, W, y, x);
                if (t.y_low < 0) continue;
                float v1 = elem<T>::to_f32(d[t.y_low * W + t.x_low]);
                float v2 = elem<T>::to_f32(d[t.y_low * W + t.x_high]);
                float v3 = elem<T>::to_f32(d[t.y_high * W + t.x_low]);
                float v4 = elem<T>::to_f32(d[t.y_high * W + t.x_high]);
                float val = t.w1 * v1 + t.w2 * v2 + t.w3 * v3 + t.w4 * v4;
                acc += val;
            }
        }
        acc /= g.count;
        out[index] = elem<T>::from_f32(acc);
    }
}

// ---------------------------------------------------------------------------------------------
// ROIAlign backward (fp32, atomics).  ROIAlign_cuda.cu:201-278.
__global__ void roi_align_bwd_nhwc_kernel(const float* __restrict__ grad, const float* __restrict__ rois, int C, int H,
                                          int W, int ph, int pw, float scale, int sampling_ratio,
                                          float* __restrict__ gfeat) {
    const int bin = blockIdx.x;
    const int n = bin / (ph * pw);
    const int p = (bin / pw) % ph;
    const int q = bin % pw;
    const RoiGeom g = roi_geom(rois + 5 * n, scale, ph, pw, sampling_ratio);
    float* base = gfeat + (size_t)g.batch * H * W * C;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float gtop = grad[(size_t)bin * C + c];
        for (int iy = 0; iy < g.grid_h; ++iy) {
            const float y = sample_y(g, p, iy);
            for (int ix = 0; ix < g.grid_w; ++ix) {
                const float x = sample_x(g, q, ix);
                const Tap t = bilinear_tap(H, W, y, x);
                if (t.y_low < 0) continue;
                float g1 = gtop * t.w1 / g.count;
                float g2 = gtop * t.w2 / g.count;
                float g3 = gtop * t.w3 / g.count;
                float g4 = gtop * t.w4 / g.count;
                atomicAdd(base + ((size_t)t.y_low * W + t.x_low) * C + c, g1);
                atomicAdd(base + ((size_t)t.y_low * W + t.x_high) * C + c, g2);
                atomicAdd(base + ((size_t)t.y_high * W + t.x_low) * C + c, g3);
                atomicAdd(base + ((size_t)t.y_high * W + t.x_high) * C + c, g4);
            }
        }
    }
}

__global__ void roi_align_bwd_nchw_kernel(const float* __restrict__ grad, const float* __restrict__ rois,
                                          long long total, int C, int H, int W, int ph, int pw, float scale,
                                          int sampling_ratio, float* __restrict__ gfeat) {
    for (long long index = (long long)blockIdx.x * blockDim.x + threadIdx.x; index < total;
         index += (long long)blockDim.x * gridDim.x) {
        const int q = (int)(index % pw);
        const int p = (int)((index / pw) % ph);
        const int c = (int)((index / pw / ph) % C);
        const int n = (int)(index / pw / ph / C);
        const RoiGeom g = roi_geom(rois + 5 * n, scale, ph, pw, sampling_ratio);
        float* d = gfeat + ((size_t)g.batch * C + c) * H * W;
        const float gtop = grad[index];
        for (int iy = 0; iy < g.grid_h; ++iy) {
            const float y = sample_y(g, p, iy);
            for (int ix = 0; ix < g.grid_w; ++ix) {
                const float x = sample_x(g, q, ix);
                const Tap t = bilinear_tap(H, W, y, x);
                if (t.y_low < 0) continue;
                atomicAdd(d + t.y_low * W + t.x_low, gtop * t.w1 / g.count);
                atomicAdd(d + t.y_low * W + t.x_high, gtop * t.w2 / g.count);
                atomicAdd(d + t.y_high * W + t.x_low, gtop * t.w3 / g.count);
                atomicAdd(d + t.y_high * W + t.x_high, gtop * t.w4 / g.count);
            }
        }
    }
}

// ROIAlign backward as a GATHER: one output pixel (frame b, row Y, column X) sums, in a fixed order (rois ascending, then bin row,
// sample row, bin column, sample column), every sample whose bilinear footprint touches it -- the same products gtop * w / count as
// the scatter form above, no atomics, so the gradient is bit-reproducible run to run.  The footprint is separable: a sample row y
// touches pixel row Y with weight hy (y_low == Y) and / or ly (y_high == Y) of bilinear_tap; likewise for columns.
__device__ __forceinline__ float axis_weight(int size, float v, int P) {
    if (v < -1.0f || v > (float)size) return 0.f;          // void sample (bilinear_tap's first test, per axis)
    if (v <= 0) v = 0;
    int lo = (int)v, hi;
    if (lo >= size - 1) { hi = lo = size - 1; v = (float)lo; } else { hi = lo + 1; }
    const float l = v - (float)lo, h = 1.f - l;
    return (lo == P ? h : 0.f) + (hi == P && hi != lo ? l : 0.f);
}

// One workgroup per feature cell, lanes along channels.  The bilinear weights of a roi depend on the cell only, not on the channel:
// the first lanes compute the roi's row weights wy[bin row x sample row] and column weights wx[...] ONCE into LDS (every branch
// here is workgroup-uniform -- it depends on blockIdx and the roi list only -- so the barriers inside the roi loop are legal), then
// every lane walks the non-zero pairs.  (Each lane evaluating the weights itself: 205 us per call in the C4 step, VALU-bound.)
constexpr int RBG_MAXS = 128;                       // samples per axis the LDS tables hold (pooled size x sampling grid); larger: per-lane path
constexpr int RBG_RB = 8;                           // rois per round of the listed walk: their weight tables are built together (two barriers per round, not per roi)
constexpr int RBG_LIST = 512;                       // rois per cell the LDS candidate list holds; a cell that more rois touch walks all K headers
// a roi can reach the cell (frame b, row Y, column X): samples lie inside [start, start + max(extent, 1)]; a pixel more than one cell
// away on either axis gets nothing
__device__ __forceinline__ bool roi_touches(const float* r, int b, int Y, int X, float scale, int ph, int pw, int sampling_ratio) {
    if ((int)r[0] != b) return false;
    const RoiGeom g = roi_geom(r, scale, ph, pw, sampling_ratio);
    return !(g.start_h > (float)(Y + 1) || g.start_h + g.bin_h * (float)ph < (float)(Y - 1) ||
             g.start_w > (float)(X + 1) || g.start_w + g.bin_w * (float)pw < (float)(X - 1));
}
// (Measured and not kept, round 4: one workgroup per ROW of cells -- the header scan and candidate list shared by the row's cells, the
// cells walked one after the other -- 1.70 -> 2.57 ms at K = 1080, 0.56 -> 0.95 at K = 360: the scan is not what bounds the call, and
// 25 cells in sequence per workgroup cost more latency than 24 saved scans.  tools/roi_bwd_bench.py.)
template <int V>
__global__ void roi_align_bwd_gather_nhwc_kernel(const float* __restrict__ grad, const float* __restrict__ rois, int K, int C, int H, int W,
                                                 int ph, int pw, float scale, int sampling_ratio, float* __restrict__ gfeat) {
    __shared__ float wy_s[RBG_RB][RBG_MAXS], wx_s[RBG_RB][RBG_MAXS];
    __shared__ int rng_s[RBG_RB][4];                  // per roi of the round: first / last sample row, first / last sample column with a non-zero weight
    __shared__ int list_s[RBG_LIST];
    __shared__ float box_s[RBG_LIST][5];              // the listed rois' rows: the rounds below read their geometry from LDS, not through a chain of global loads
    __shared__ int nlist_s;
    const int pix = blockIdx.x;
    const int X = pix % W, Y = (pix / W) % H, b = pix / (W * H);
    const int tid = threadIdx.x, nthr = blockDim.x;
    // The rois that touch this cell, ascending, found by the first wavefront 64 headers at a time (ballot + prefix count keep the
    // order): a training step pools hundreds of rois (8 clips x 15 tubes x 9 frames = 1080) of which a cell's frame holds a few, and
    // every cell walking all K headers one after the other was 2.8 ms per call there (profiles/r04: 9 % of the C4 step at 8 clips).
    if (tid < 64) {
        int cnt = 0;
#pragma unroll 4
        for (int base = 0; base < K; base += 64) {    // (unrolled: the header loads of four rounds are in flight together)
            const int n = base + tid;
            float r5[5] = {-1.f, 0.f, 0.f, 0.f, 0.f};
            if (n < K) {
#pragma unroll
                for (int j = 0; j < 5; ++j) r5[j] = rois[5 * n + j];
            }
            const bool hit = n < K && roi_touches(r5, b, Y, X, scale, ph, pw, sampling_ratio);
            const unsigned long long m = __ballot(hit);
            if (hit) {
                const int pos = cnt + __builtin_popcountll(m & ((1ull << tid) - 1ull));
                if (pos < RBG_LIST) {
                    list_s[pos] = n;
#pragma unroll
                    for (int j = 0; j < 5; ++j) box_s[pos][j] = r5[j];
                }
            }
            cnt += __builtin_popcountll(m);
        }
        if (tid == 0) nlist_s = cnt;
    }
    __syncthreads();
    const int nlist = nlist_s;
    const bool listed = nlist <= RBG_LIST;          // workgroup-uniform
    const int nwalk = listed ? nlist : K;
    constexpr int MAXCV = 4;                        // channel vectors per lane and pass (832 channels: one pass of 208 lanes x 4)
    for (int cbase = 0; cbase < C; cbase += nthr * V * MAXCV) {
    float acc[MAXCV][V];
#pragma unroll
    for (int k = 0; k < MAXCV; ++k)
#pragma unroll
        for (int i = 0; i < V; ++i) acc[k][i] = 0.f;
    // Listed walk, RBG_RB rois per round.  A sample row's weight for cell row Y is a function of the sample coordinate alone, and the
    // coordinates grow with the sample index: the samples that reach this cell are a CONTIGUOUS index range per axis (2-3 of the 7 x
    // grid samples).  The round's first phase writes every roi's two weight tables and finds the ranges (LDS min / max: order-
    // independent), the second walks range x range instead of all (7 grid)^2 table entries -- same products, same order (rois
    // ascending, then sample row, then sample column).
    for (int w0 = 0; listed && w0 < nlist; w0 += RBG_RB) {
        const int nr = min(RBG_RB, nlist - w0);
        if (tid < RBG_RB * 4) rng_s[tid >> 2][tid & 3] = (tid & 1) ? -1 : 0x7fffffff;
        __syncthreads();
        bool small = true;                            // (workgroup-uniform: every roi of the round fits the tables)
        for (int r = 0; r < nr; ++r) {
            const RoiGeom g = roi_geom(box_s[w0 + r], scale, ph, pw, sampling_ratio);
            const int SY = ph * g.grid_h, SX = pw * g.grid_w;
            if (SY > RBG_MAXS || SX > RBG_MAXS) { small = false; continue; }
            for (int i = tid; i < SY + SX; i += nthr) {
                if (i < SY) {
                    const float w = axis_weight(H, sample_y(g, i / g.grid_h, i % g.grid_h), Y);
                    wy_s[r][i] = w;
                    if (w != 0.f) { atomicMin(&rng_s[r][0], i); atomicMax(&rng_s[r][1], i); }
                } else {
                    const int j = i - SY;
                    const float w = axis_weight(W, sample_x(g, j / g.grid_w, j % g.grid_w), X);
                    wx_s[r][j] = w;
                    if (w != 0.f) { atomicMin(&rng_s[r][2], j); atomicMax(&rng_s[r][3], j); }
                }
            }
        }
        __syncthreads();
        for (int r = 0; r < nr; ++r) {
            const int n = list_s[w0 + r];
            const RoiGeom g = roi_geom(box_s[w0 + r], scale, ph, pw, sampling_ratio);
            const int SY = ph * g.grid_h, SX = pw * g.grid_w;
            const bool tables = SY <= RBG_MAXS && SX <= RBG_MAXS;
            const int y0 = tables ? rng_s[r][0] : 0, y1 = tables ? rng_s[r][1] : SY - 1;
            const int x0 = tables ? rng_s[r][2] : 0, x1 = tables ? rng_s[r][3] : SX - 1;
            for (int sy = y0; sy <= y1; ++sy) {
                const int p = sy / g.grid_h;
                const float wy = tables ? wy_s[r][sy] : axis_weight(H, sample_y(g, p, sy % g.grid_h), Y);
                if (wy == 0.f) continue;
                for (int sx = x0; sx <= x1; ++sx) {
                    const int q = sx / g.grid_w;
                    const float wx = tables ? wx_s[r][sx] : axis_weight(W, sample_x(g, q, sx % g.grid_w), X);
                    if (wx == 0.f) continue;
                    const float w = wy * wx;
#pragma unroll
                    for (int k = 0; k < MAXCV; ++k) {
                        const int c = cbase + (tid + k * nthr) * V;
                        if (c < C) {
                            float gt[V];
                            VecIO<float, V>::load(grad + ((size_t)(n * ph + p) * pw + q) * C + c, gt);
#pragma unroll
                            for (int i = 0; i < V; ++i) acc[k][i] += gt[i] * w / g.count;
                        }
                    }
                }
            }
        }
        (void)small;
        __syncthreads();                              // the tables are rewritten in the next round
    }
    // a cell that more rois touch than the list holds: every header, one roi per barrier pair (round 3's walk)
    for (int w_ = 0; !listed && w_ < nwalk; ++w_) {
        const int n = w_;
        if (!roi_touches(rois + 5 * n, b, Y, X, scale, ph, pw, sampling_ratio)) continue;
        const RoiGeom g = roi_geom(rois + 5 * n, scale, ph, pw, sampling_ratio);
        const int SY = ph * g.grid_h, SX = pw * g.grid_w;
        const bool tables = SY <= RBG_MAXS && SX <= RBG_MAXS;
        if (tables) {
            for (int i = tid; i < SY + SX; i += nthr) {
                if (i < SY) wy_s[0][i] = axis_weight(H, sample_y(g, i / g.grid_h, i % g.grid_h), Y);
                else wx_s[0][i - SY] = axis_weight(W, sample_x(g, (i - SY) / g.grid_w, (i - SY) % g.grid_w), X);
            }
            __syncthreads();
        }
        for (int p = 0; p < ph; ++p)
            for (int iy = 0; iy < g.grid_h; ++iy) {
                const float wy = tables ? wy_s[0][p * g.grid_h + iy] : axis_weight(H, sample_y(g, p, iy), Y);
                if (wy == 0.f) continue;
                // (a sample whose OTHER coordinate is void contributes nothing: its column weight below is 0)
                for (int q = 0; q < pw; ++q)
                    for (int ix = 0; ix < g.grid_w; ++ix) {
                        const float wx = tables ? wx_s[0][q * g.grid_w + ix] : axis_weight(W, sample_x(g, q, ix), X);
                        if (wx == 0.f) continue;
                        const float w = wy * wx;
#pragma unroll
                        for (int k = 0; k < MAXCV; ++k) {
                            const int c = cbase + (tid + k * nthr) * V;
                            if (c < C) {
                                float gt[V];
                                VecIO<float, V>::load(grad + ((size_t)(n * ph + p) * pw + q) * C + c, gt);
#pragma unroll
                                for (int i = 0; i < V; ++i) acc[k][i] += gt[i] * w / g.count;
                            }
                        }
                    }
            }
        if (tables) __syncthreads();                // the tables are rewritten for the next roi
    }
#pragma unroll
    for (int k = 0; k < MAXCV; ++k) {
        const int c = cbase + (tid + k * nthr) * V;
        if (c < C) VecIO<float, V>::store(gfeat + (size_t)pix * C + c, acc[k]);
    }
    }
}

__global__ void roi_align_bwd_gather_nchw_kernel(const float* __restrict__ grad, const float* __restrict__ rois, int K, long long total, int C,
                                                 int H, int W, int ph, int pw, float scale, int sampling_ratio, float* __restrict__ gfeat) {
    for (long long index = (long long)blockIdx.x * blockDim.x + threadIdx.x; index < total; index += (long long)blockDim.x * gridDim.x) {
        const int X = (int)(index % W), Y = (int)((index / W) % H);
        const int c = (int)((index / W / H) % C), b = (int)(index / W / H / C);
        float acc = 0.f;
        for (int n = 0; n < K; ++n) {
            if ((int)rois[5 * n] != b) continue;
            const RoiGeom g = roi_geom(rois + 5 * n, scale, ph, pw, sampling_ratio);
            if (g.start_h > (float)(Y + 1) || g.start_h + g.bin_h * (float)ph < (float)(Y - 1) ||
                g.start_w > (float)(X + 1) || g.start_w + g.bin_w * (float)pw < (float)(X - 1)) continue;
            const float* gr = grad + ((size_t)n * C + c) * ph * pw;
            for (int p = 0; p < ph; ++p)
                for (int iy = 0; iy < g.grid_h; ++iy) {
                    const float wy = axis_weight(H, sample_y(g, p, iy), Y);
                    if (wy == 0.f) continue;
                    for (int q = 0; q < pw; ++q)
                        for (int ix = 0; ix < g.grid_w; ++ix) {
                            const float wx = axis_weight(W, sample_x(g, q, ix), X);
                            if (wx == 0.f) continue;
                            acc += gr[p * pw + q] * (wy * wx) / g.count;
                        }
                }
        }
        gfeat[index] = acc;
    }
}

// ---------------------------------------------------------------------------------------------
// ROIPool.  ROIPool_cuda.cu:40-132.
struct PoolBin { int batch, hstart, hend, wstart, wend; };

__device__ __forceinline__ PoolBin roi_pool_bin(const float* r, float scale, int ph, int pw, int p, int q, int H, int W) {
    PoolBin b;
    b.batch = (int)r[0];
    int sw = (int)roundf(r[1] * scale);
    int sh = (int)roundf(r[2] * scale);
    int ew = (int)roundf(r[3] * scale);
    int eh = (int)roundf(r[4] * scale);
    int rw = max(ew - sw + 1, 1);
    int rh = max(eh - sh + 1, 1);
    float bin_h = (float)rh / (float)ph;
    float bin_w = (float)rw / (float)pw;
    int hstart = (int)floorf((float)p * bin_h);
    int wstart = (int)floorf((float)q * bin_w);
    int hend = (int)ceilf((float)(p + 1) * bin_h);
    int wend = (int)ceilf((float)(q + 1) * bin_w);
    b.hstart = min(max(hstart + sh, 0), H);
    b.hend = min(max(hend + sh, 0), H);
    b.wstart = min(max(wstart + sw, 0), W);
    b.wend = min(max(wend + sw, 0), W);
    return b;
}

template <typename T>
__global__ void roi_pool_fwd_nhwc_kernel(const T* __restrict__ feat, const float* __restrict__ rois, int C, int H, int W,
                                         int ph, int pw, float scale, T* __restrict__ out, int32_t* __restrict__ argmax) {
    const int bin = blockIdx.x;
    const int n = bin / (ph * pw);
    const int p = (bin / pw) % ph;
    const int q = bin % pw;
    const PoolBin b = roi_pool_bin(rois + 5 * n, scale, ph, pw, p, q, H, W);
    const bool empty = (b.hend <= b.hstart) || (b.wend <= b.wstart);
    const T* base = feat + (size_t)b.batch * H * W * C;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float maxval = empty ? 0.f : -FLT_MAX;
        int maxidx = -1;
        for (int h = b.hstart; h < b.hend; ++h)
            for (int w = b.wstart; w < b.wend; ++w) {
                float v = elem<T>::to_f32(base[((size_t)h * W + w) * C + c]);
                if (v > maxval) { maxval = v; maxidx = h * W + w; }
            }
        out[(size_t)bin * C + c] = elem<T>::from_f32(maxval);
        argmax[(size_t)bin * C + c] = maxidx;
    }
}

template <typename T>
__global__ void roi_pool_fwd_nchw_kernel(const T* __restrict__ feat, const float* __restrict__ rois, long long total,
                                         int C, int H, int W, int ph, int pw, float scale, T* __restrict__ out,
                                         int32_t* __restrict__ argmax) {
    for (long long index = (long long)blockIdx.x * blockDim.x + threadIdx.x; index < total;
         index += (long long)blockDim.x * gridDim.x) {
        const int q = (int)(index % pw);
        const int p = (int)((index / pw) % ph);
        const int c = (int)((index / pw / ph) % C);
        const int n = (int)(index / pw / ph / C);
        const PoolBin b = roi_pool_bin(rois + 5 * n, scale, ph, pw, p, q, H, W);
        const bool empty = (b.hend <= b.hstart) || (b.wend <= b.wstart);
        const T* d = feat + ((size_t)b.batch * C + c) * H * W;
        float maxval = empty ? 0.f : -FLT_MAX;
        int maxidx = -1;
        for (int h = b.hstart; h < b.hend; ++h)
            for (int w = b.wstart; w < b.wend; ++w) {
                float v = elem<T>::to_f32(d[h * W + w]);
                if (v > maxval) { maxval = v; maxidx = h * W + w; }
            }
        out[index] = elem<T>::from_f32(maxval);
        argmax[index] = maxidx;
    }
}

// layout: 0 = NCHW (index = ((n*C+c)*ph+p)*pw+q), 1 = NHWC (index = ((n*ph+p)*pw+q)*C+c)
__global__ void roi_pool_bwd_kernel(const float* __restrict__ grad, const int32_t* __restrict__ argmax,
                                    const float* __restrict__ rois, long long total, int layout, int C, int H, int W,
                                    int ph, int pw, float* __restrict__ gfeat) {
    for (long long index = (long long)blockIdx.x * blockDim.x + threadIdx.x; index < total;
         index += (long long)blockDim.x * gridDim.x) {
        int c, n;
        if (layout == STEP_NCHW) {
            c = (int)((index / pw / ph) % C);
            n = (int)(index / pw / ph / C);
        } else {
            c = (int)(index % C);
            n = (int)(index / C / pw / ph);
        }
        const int a = argmax[index];
        if (a == -1) continue;
        const int batch = (int)rois[5 * n];
        if (layout == STEP_NCHW)
            atomicAdd(gfeat + ((size_t)batch * C + c) * H * W + a, grad[index]);
        else
            atomicAdd(gfeat + ((size_t)batch * H * W + a) * C + c, grad[index]);
    }
}

static inline int lanes_for(int items) {
    int t = 64;
    while (t < items && t < 256) t <<= 1;
    return t;
}
static inline unsigned flat_grid(long long total, int block) {
    long long g = ceil_div64(total, block);
    if (g > 8192) g = 8192;  // grid-stride the rest: 256 CUs x 8 workgroups x 4
    return (unsigned)g;
}

template <typename T>
static int roi_align_forward_t(const void* feat, int layout, const float* rois, int K, int C, int H, int W, int ph,
                               int pw, float scale, int sr, void* out, step_stream_t stream, int tsl = 0, int tall = 0) {
    constexpr int V = elem<T>::VEC;
    if (layout == STEP_NHWC) {
        if (C % V == 0 && ((uintptr_t)feat % 16) == 0 && ((uintptr_t)out % 16) == 0) {
            STEP_LAUNCH((roi_align_fwd_nhwc_kernel<T, V>), dim3(K * ph * pw), dim3(lanes_for(C / V)), stream,
                        (const T*)feat, rois, C, H, W, ph, pw, scale, sr, (T*)out, tsl, tall);
        } else {
            STEP_LAUNCH((roi_align_fwd_nhwc_kernel<T, 1>), dim3(K * ph * pw), dim3(lanes_for(C)), stream,
                        (const T*)feat, rois, C, H, W, ph, pw, scale, sr, (T*)out, tsl, tall);
        }
    } else {
        long long total = (long long)K * C * ph * pw;
        STEP_LAUNCH((roi_align_fwd_nchw_kernel<T>), dim3(flat_grid(total, 256)), dim3(256), stream, (const T*)feat,
                    rois, total, C, H, W, ph, pw, scale, sr, (T*)out);
    }
    return STEP_LAUNCH_CHECK();
}

template <typename T>
static int roi_pool_forward_t(const void* feat, int layout, const float* rois, int K, int C, int H, int W, int ph,
                              int pw, float scale, void* out, int32_t* argmax, step_stream_t stream) {
    if (layout == STEP_NHWC) {
        STEP_LAUNCH((roi_pool_fwd_nhwc_kernel<T>), dim3(K * ph * pw), dim3(lanes_for(C)), stream, (const T*)feat, rois,
                    C, H, W, ph, pw, scale, (T*)out, argmax);
    } else {
        long long total = (long long)K * C * ph * pw;
        STEP_LAUNCH((roi_pool_fwd_nchw_kernel<T>), dim3(flat_grid(total, 256)), dim3(256), stream, (const T*)feat,
                    rois, total, C, H, W, ph, pw, scale, (T*)out, argmax);
    }
    return STEP_LAUNCH_CHECK();
}

}  // namespace step

using namespace step;

extern "C" {

int step_roi_align_forward(const void* feat, int dtype, int layout, const float* rois, int K, int B, int C, int H,
                           int W, int ph, int pw, float scale, int sr, void* out, step_stream_t stream) {
    if (K < 0 || B < 0 || C <= 0 || H <= 0 || W <= 0 || ph <= 0 || pw <= 0) return STEP_E_SHAPE;
    if (layout != STEP_NCHW && layout != STEP_NHWC) return STEP_E_UNSUPPORTED;
    if (K == 0) return STEP_OK;
    if (!feat || !rois || !out) return STEP_E_NULL;
    switch (dtype) {
        case STEP_F32: return roi_align_forward_t<float>(feat, layout, rois, K, C, H, W, ph, pw, scale, sr, out, stream);
        case STEP_BF16: return roi_align_forward_t<bf16_t>(feat, layout, rois, K, C, H, W, ph, pw, scale, sr, out, stream);
        case STEP_F16: return roi_align_forward_t<f16_t>(feat, layout, rois, K, C, H, W, ph, pw, scale, sr, out, stream);
    }
    return STEP_E_DTYPE;
}

int step_roi_align_tubes_forward(const void* feat, int dtype, const float* rois, int K, int B, int T_all, int T, int C, int H, int W,
                                 int ph, int pw, float scale, int sr, void* out, step_stream_t stream) {
    if (K < 0 || B < 0 || C <= 0 || H <= 0 || W <= 0 || ph <= 0 || pw <= 0 || T <= 0 || T_all < T) return STEP_E_SHAPE;
    if (K == 0) return STEP_OK;
    if (!feat || !rois || !out) return STEP_E_NULL;
    switch (dtype) {
        case STEP_F32: return roi_align_forward_t<float>(feat, STEP_NHWC, rois, K, C, H, W, ph, pw, scale, sr, out, stream, T, T_all);
        case STEP_BF16: return roi_align_forward_t<bf16_t>(feat, STEP_NHWC, rois, K, C, H, W, ph, pw, scale, sr, out, stream, T, T_all);
        case STEP_F16: return roi_align_forward_t<f16_t>(feat, STEP_NHWC, rois, K, C, H, W, ph, pw, scale, sr, out, stream, T, T_all);
    }
    return STEP_E_DTYPE;
}

int step_roi_align_backward(const float* grad, int layout, const float* rois, int K, int B, int C, int H, int W,
                            int ph, int pw, float scale, int sr, int mode, float* gfeat, step_stream_t stream) {
    if (K < 0 || B < 0 || C <= 0 || H <= 0 || W <= 0 || ph <= 0 || pw <= 0) return STEP_E_SHAPE;
    if (layout != STEP_NCHW && layout != STEP_NHWC) return STEP_E_UNSUPPORTED;
    if (mode != STEP_ROI_BWD_GATHER && mode != STEP_ROI_BWD_ATOMIC) return STEP_E_UNSUPPORTED;
    if (B == 0) return STEP_OK;
    if (!gfeat) return STEP_E_NULL;
    if (K > 0 && mode == STEP_ROI_BWD_GATHER) {
        // the deterministic form: every element of grad_feat is written by its own gather (no clear, no atomics)
        if (!grad || !rois) return STEP_E_NULL;
        if (layout == STEP_NHWC) {
            const bool v4 = (C % 4) == 0 && ((uintptr_t)grad % 16) == 0 && ((uintptr_t)gfeat % 16) == 0;
            const int threads = v4 ? lanes_for(C / 4) : lanes_for(C);
            if (v4) STEP_LAUNCH((roi_align_bwd_gather_nhwc_kernel<4>), dim3((unsigned)(B * H * W)), dim3(threads), stream, grad, rois, K, C, H, W, ph, pw,
                                scale, sr, gfeat);
            else STEP_LAUNCH((roi_align_bwd_gather_nhwc_kernel<1>), dim3((unsigned)(B * H * W)), dim3(threads), stream, grad, rois, K, C, H, W, ph, pw,
                             scale, sr, gfeat);
        } else {
            const long long total = (long long)B * C * H * W;
            STEP_LAUNCH((roi_align_bwd_gather_nchw_kernel), dim3(flat_grid(total, 256)), dim3(256), stream, grad, rois, K, total, C, H, W, ph, pw,
                        scale, sr, gfeat);
        }
        return STEP_LAUNCH_CHECK();
    }
    int rc = (int)hipMemsetAsync(gfeat, 0, sizeof(float) * (size_t)B * C * H * W, (hipStream_t)stream);
    if (rc) return rc;
    if (K == 0) return STEP_OK;
    if (!grad || !rois) return STEP_E_NULL;
    if (layout == STEP_NHWC) {
        STEP_LAUNCH((roi_align_bwd_nhwc_kernel), dim3(K * ph * pw), dim3(lanes_for(C)), stream, grad, rois, C, H, W, ph,
                    pw, scale, sr, gfeat);
    } else {
        long long total = (long long)K * C * ph * pw;
        STEP_LAUNCH((roi_align_bwd_nchw_kernel), dim3(flat_grid(total, 256)), dim3(256), stream, grad, rois, total, C,
                    H, W, ph, pw, scale, sr, gfeat);
    }
    return STEP_LAUNCH_CHECK();
}

int step_roi_pool_forward(const void* feat, int dtype, int layout, const float* rois, int K, int B, int C, int H,
                          int W, int ph, int pw, float scale, void* out, int32_t* argmax, step_stream_t stream) {
    if (K < 0 || B < 0 || C <= 0 || H <= 0 || W <= 0 || ph <= 0 || pw <= 0) return STEP_E_SHAPE;
    if (layout != STEP_NCHW && layout != STEP_NHWC) return STEP_E_UNSUPPORTED;
    if (K == 0) return STEP_OK;
    if (!feat || !rois || !out || !argmax) return STEP_E_NULL;
    switch (dtype) {
        case STEP_F32: return roi_pool_forward_t<float>(feat, layout, rois, K, C, H, W, ph, pw, scale, out, argmax, stream);
        case STEP_BF16: return roi_pool_forward_t<bf16_t>(feat, layout, rois, K, C, H, W, ph, pw, scale, out, argmax, stream);
        case STEP_F16: return roi_pool_forward_t<f16_t>(feat, layout, rois, K, C, H, W, ph, pw, scale, out, argmax, stream);
    }
    return STEP_E_DTYPE;
}

int step_roi_pool_backward(const float* grad, const int32_t* argmax, int layout, const float* rois, int K, int B,
                           int C, int H, int W, int ph, int pw, float* gfeat, step_stream_t stream) {
    if (K < 0 || B < 0 || C <= 0 || H <= 0 || W <= 0 || ph <= 0 || pw <= 0) return STEP_E_SHAPE;
    if (layout != STEP_NCHW && layout != STEP_NHWC) return STEP_E_UNSUPPORTED;
    if (B == 0) return STEP_OK;
    if (!gfeat) return STEP_E_NULL;
    int rc = (int)hipMemsetAsync(gfeat, 0, sizeof(float) * (size_t)B * C * H * W, (hipStream_t)stream);
    if (rc) return rc;
    if (K == 0) return STEP_OK;
    if (!grad || !rois || !argmax) return STEP_E_NULL;
    long long total = (long long)K * C * ph * pw;
    STEP_LAUNCH((roi_pool_bwd_kernel), dim3(flat_grid(total, 256)), dim3(256), stream, grad, argmax, rois, total,
                layout, C, H, W, ph, pw, gfeat);
    return STEP_LAUNCH_CHECK();
}

}  // extern "C"
