// step_amd/csrc/nms.hip -- batched greedy NMS for gfx950, bit-exact with the reference CPU op.
//
// Replaces  external/maskrcnn_benchmark/csrc/cpu/nms_cpu.cpp:29-89 (the operator every reference
// script actually reaches: test.py:158-161,192 / train.py:513-515,547 / demo.py:124-126,158 move the
// boxes to the CPU first) and csrc/cuda/nms.cu:47-155 (64x64 bit-mask tiles + a serial host
// reduction behind a blocking D2H copy).
//
// STEP calls nms once per (step, clip, class) on <= 34..109 middle-frame boxes: 180*B tiny serial
// CPU calls per batch.  Here all groups go in ONE launch:
//   * kmax <= 64  : one 64-lane wavefront per group, boxes live in registers, the greedy scan is a
//                   loop of cross-lane broadcasts -- no LDS, no global scratch, no host round trip;
//   * kmax  > 64  : one 256-thread workgroup per group with a global scratch rank table.
// Arithmetic: "+1" areas, IoU = inter / (area_i + area_j - inter) with every operation rounded
// separately (__f*_rn: no FMA contraction), suppress when IoU >= threshold (nms_cpu.cpp:84 -- the
// CUDA op uses '>', nms.cu:84; parity target is the CPU op), ties in score broken by lower index.
#include "common.h"

#pragma clang fp contract(off)

namespace step {

__device__ __forceinline__ float box_area(float x1, float y1, float x2, float y2) {
    return __fmul_rn(__fadd_rn(__fsub_rn(x2, x1), 1.f), __fadd_rn(__fsub_rn(y2, y1), 1.f));  // nms_cpu.cpp:46
}
// fp64 boxes (the reference dispatches AT_DISPATCH_FLOATING_TYPES, nms_cpu.cpp:95): the same operations in double, each rounded
// separately (this file is compiled with fp contraction off); the threshold stays the operator's float argument, promoted in
// `ovr >= threshold` as in nms_cpu_kernel<double>.
__device__ __forceinline__ double box_area(double x1, double y1, double x2, double y2) {
    const double w = (x2 - x1) + 1.0, h = (y2 - y1) + 1.0;
    return w * h;
}
__device__ __forceinline__ bool iou_ge(double ix1, double iy1, double ix2, double iy2, double iarea, double jx1, double jy1,
                                       double jx2, double jy2, double jarea, float thr) {
    const double xx1 = fmax(ix1, jx1), yy1 = fmax(iy1, jy1);
    const double xx2 = fmin(ix2, jx2), yy2 = fmin(iy2, jy2);
    const double w = fmax(0.0, (xx2 - xx1) + 1.0);
    const double h = fmax(0.0, (yy2 - yy1) + 1.0);
    const double inter = w * h;
    const double ovr = inter / ((iarea + jarea) - inter);
    return ovr >= (double)thr;
}

// i = the kept (higher score) box, j = the candidate.  nms_cpu.cpp:73-84
__device__ __forceinline__ bool iou_ge(float ix1, float iy1, float ix2, float iy2, float iarea, float jx1, float jy1,
                                       float jx2, float jy2, float jarea, float thr) {
    float xx1 = fmaxf(ix1, jx1), yy1 = fmaxf(iy1, jy1);
    float xx2 = fminf(ix2, jx2), yy2 = fminf(iy2, jy2);
    float w = fmaxf(0.f, __fadd_rn(__fsub_rn(xx2, xx1), 1.f));
    float h = fmaxf(0.f, __fadd_rn(__fsub_rn(yy2, yy1), 1.f));
    float inter = __fmul_rn(w, h);
    float ovr = __fdiv_rn(inter, __fsub_rn(__fadd_rn(iarea, jarea), inter));
    return ovr >= thr;
}

// Score order: descending, ties by lower index.  NaN scores sort FIRST (torch's sort, which the reference uses at
// nms_cpu.cpp:50, treats NaN as the largest value); without this rule NaNs compare false both ways, ranks collide and
// the rank -> lane lookup below would be undefined.
template <typename F>
__device__ __forceinline__ bool score_before(F sj, int j, F s, int i) {
    const bool nj = sj != sj, ni = s != s;
    if (nj || ni) return nj && (!ni || j < i);
    return sj > s || (sj == s && j < i);
}

// One wavefront per group, n <= 64.
template <typename F>
__global__ void nms_wave_kernel(const F* __restrict__ boxes, const F* __restrict__ scores,
                                const int32_t* __restrict__ counts, int kmax, float thr, uint8_t* __restrict__ keep) {
    const int g = blockIdx.x;
    const int lane = threadIdx.x;  // blockDim.x == 64
    const int n = min(counts[g], kmax);
    const bool valid = lane < n;
    F x1 = 0, y1 = 0, x2 = 0, y2 = 0, s = 0;
    if (valid) {
        const F* b = boxes + ((size_t)g * kmax + lane) * 4;
        x1 = b[0]; y1 = b[1]; x2 = b[2]; y2 = b[3];
        s = scores[(size_t)g * kmax + lane];
    }
    const F area = box_area(x1, y1, x2, y2);
    // stable descending rank (ties: lower index first)
    int rank = 0;
    for (int j = 0; j < n; ++j) {
        F sj = __shfl(s, j);
        rank += score_before(sj, j, s, lane) ? 1 : 0;
    }
    bool suppressed = false;
    for (int r = 0; r < n; ++r) {
        unsigned long long m = __ballot(valid && rank == r);
        int i = __builtin_ctzll(m);  // exactly one lane has rank r
        unsigned long long sm = __ballot(suppressed);
        F bx1 = __shfl(x1, i), by1 = __shfl(y1, i), bx2 = __shfl(x2, i), by2 = __shfl(y2, i);
        F barea = __shfl(area, i);
        if ((sm >> i) & 1ull) continue;  // wave-uniform
        if (valid && !suppressed && rank > r && iou_ge(bx1, by1, bx2, by2, barea, x1, y1, x2, y2, area, thr))
            suppressed = true;
    }
    if (lane < kmax) keep[(size_t)g * kmax + lane] = (valid && !suppressed) ? 1 : 0;
}

// The evaluation loop of test.py:157-198 for one refinement iteration as ONE launch: wavefront (b, c) takes clip b's tubes (slot j =
// lane), masks them with score > conf (test.py:180), clamps their middle-frame boxes the way valid_tubes does (tube_utils.py:59-92:
// clip to [0, width] x [0, height], boxes under 3 px become the whole frame) and runs the greedy NMS among the masked ones -- the
// reference compacts them first, keeping their order, so score ties still go to the lower original slot.  keep[b][c][j] marks the
// survivors at their ORIGINAL slots.  n <= 64 tubes per clip.
__global__ void detect_nms_wave_kernel(const float* __restrict__ prob, long long prob_stride, int NC, const float* __restrict__ loc,
                                       long long loc_stride, const int32_t* __restrict__ tube_start, const int32_t* __restrict__ tube_count,
                                       int kmax, float conf, float thr, float width, float height, uint8_t* __restrict__ keep,
                                       float* __restrict__ boxes_out) {
    const int b = blockIdx.x / NC, c = blockIdx.x % NC;
    const int lane = threadIdx.x;
    const int n = min(tube_count[b], kmax);
    const long long tube = (long long)tube_start[b] + lane;
    float x1 = 0.f, y1 = 0.f, x2 = 0.f, y2 = 0.f, s = 0.f;
    bool valid = false;
    if (lane < n) {
        const float* bx = loc + tube * loc_stride;
        x1 = fmaxf(0.f, bx[0]); y1 = fmaxf(0.f, bx[1]); x2 = fminf(width, bx[2]); y2 = fminf(height, bx[3]);
        // np.maximum / np.minimum PROPAGATE a NaN (fmaxf / fminf drop it), and a NaN coordinate then fails the `<` test of
        // valid_tubes: the reference turns a box with any NaN coordinate into the whole frame
        const bool nan_in = (bx[0] != bx[0]) || (bx[1] != bx[1]) || (bx[2] != bx[2]) || (bx[3] != bx[3]);
        if (nan_in || !((x1 < __fsub_rn(x2, 2.f)) && (y1 < __fsub_rn(y2, 2.f)))) { x1 = 0.f; y1 = 0.f; x2 = width; y2 = height; }
        s = prob[tube * prob_stride + c];
        valid = s > conf;
        if (c == 0 && boxes_out) {
            float* o = boxes_out + tube * 4;
            o[0] = x1; o[1] = y1; o[2] = x2; o[3] = y2;
        }
    }
    const float area = box_area(x1, y1, x2, y2);
    const unsigned long long vm = __ballot(valid);
    int rank = 0;                                              // descending score among the masked tubes, ties: lower slot first
    for (int j = 0; j < n; ++j) {
        const float sj = __shfl(s, j);
        rank += (((vm >> j) & 1ull) && score_before(sj, j, s, lane)) ? 1 : 0;
    }
    const int nv = __builtin_popcountll(vm);
    bool suppressed = false;
    for (int r = 0; r < nv; ++r) {
        const unsigned long long m = __ballot(valid && rank == r);
        const int i = __builtin_ctzll(m);                    // exactly one masked lane has rank r
        const unsigned long long sm = __ballot(suppressed);
        const float bx1 = __shfl(x1, i), by1 = __shfl(y1, i), bx2 = __shfl(x2, i), by2 = __shfl(y2, i);
        const float barea = __shfl(area, i);
        if ((sm >> i) & 1ull) continue;                       // wave-uniform
        if (valid && !suppressed && rank > r && iou_ge(bx1, by1, bx2, by2, barea, x1, y1, x2, y2, area, thr)) suppressed = true;
    }
    if (lane < kmax) keep[((size_t)b * NC + c) * kmax + lane] = (valid && !suppressed) ? 1 : 0;
}

// One 256-thread workgroup per group, any n.  scratch: order int32[G*kmax], sup uint8[G*kmax].
template <typename F>
__global__ void nms_block_kernel(const F* __restrict__ boxes, const F* __restrict__ scores,
                                 const int32_t* __restrict__ counts, int kmax, float thr, uint8_t* __restrict__ keep,
                                 int32_t* order_all, uint8_t* sup_all) {
    const int g = blockIdx.x;
    const int n = min(counts[g], kmax);
    const F* B = boxes + (size_t)g * kmax * 4;
    const F* S = scores + (size_t)g * kmax;
    int32_t* order = order_all + (size_t)g * kmax;
    uint8_t* sup = sup_all + (size_t)g * kmax;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const F s = S[i];
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            F sj = S[j];
            rank += score_before(sj, j, s, i) ? 1 : 0;
        }
        order[rank] = i;
        sup[i] = 0;
    }
    __syncthreads();
    for (int r = 0; r < n; ++r) {
        const int i = order[r];
        const bool isup = sup[i] != 0;  // uniform: written before the last barrier
        if (!isup) {
            const F ix1 = B[4 * i], iy1 = B[4 * i + 1], ix2 = B[4 * i + 2], iy2 = B[4 * i + 3];
            const F iarea = box_area(ix1, iy1, ix2, iy2);
            for (int q = r + 1 + threadIdx.x; q < n; q += blockDim.x) {
                const int j = order[q];
                if (sup[j]) continue;
                const F jx1 = B[4 * j], jy1 = B[4 * j + 1], jx2 = B[4 * j + 2], jy2 = B[4 * j + 3];
                if (iou_ge(ix1, iy1, ix2, iy2, iarea, jx1, jy1, jx2, jy2, box_area(jx1, jy1, jx2, jy2), thr)) sup[j] = 1;
            }
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < kmax; i += blockDim.x) keep[(size_t)g * kmax + i] = (i < n && !sup[i]) ? 1 : 0;
}

template <typename F>
static int nms_batched_t(const F* boxes, const F* scores, const int32_t* counts, int G, int kmax, float threshold, uint8_t* keep, void* scratch,
                         step_stream_t stream) {
    if (G < 0 || kmax < 0) return STEP_E_SHAPE;
    if (G == 0 || kmax == 0) return STEP_OK;
    if (!boxes || !scores || !counts || !keep) return STEP_E_NULL;
    if (kmax <= 64) {
        STEP_LAUNCH((nms_wave_kernel<F>), dim3(G), dim3(64), stream, boxes, scores, counts, kmax, threshold, keep);
    } else {
        if (!scratch) return STEP_E_NULL;
        int32_t* order = (int32_t*)scratch;
        uint8_t* sup = (uint8_t*)scratch + (size_t)G * kmax * 4;
        STEP_LAUNCH((nms_block_kernel<F>), dim3(G), dim3(256), stream, boxes, scores, counts, kmax, threshold, keep, order, sup);
    }
    return STEP_LAUNCH_CHECK();
}

// detect_compact_kernel -- the rows test.py:196-204 appends after the NMS of an iteration, for ALL iterations and clips in one launch: one
// 256-thread workgroup per (iteration, clip) walks its NC x kmax keep flags in row-major order (classes ascending, kept tubes in ascending
// original order: the reference's row order), numbers the set flags with a ballot prefix and writes box / [W,H,W,H], score, class, tube of
// row r to slot g * cap + r of its own fixed-capacity segment (cap = NC * kmax) -- no prefix sum across workgroups, the host reads the
// I x B counts once and slices.  (Before: nonzero + index + cat + gather + bincount, ~25 launches and two host synchronisations per step.)
struct DetectCompactParams {
    const float* boxes[STEP_DETECT_ITERS_MAX];
    const float* scores[STEP_DETECT_ITERS_MAX];
    long long score_stride[STEP_DETECT_ITERS_MAX];
    const uint8_t* keep; const int32_t* start;
    int B, NC, kmax;
    float w, h;
    float* out_boxes; float* out_scores; long long* out_cls; long long* out_tube; int32_t* counts;
};
__global__ __launch_bounds__(256) void detect_compact_kernel(DetectCompactParams p) {
    __shared__ int wsum[4];
    const int g = blockIdx.x, it = g / p.B, b = g % p.B;
    const int F = p.NC * p.kmax;
    const uint8_t* kp = p.keep + (size_t)g * F;
    const float* bx = p.boxes[it];
    const float* sc = p.scores[it];
    const long long ss = p.score_stride[it];
    const int t0 = F > 0 ? p.start[b] : 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int base = 0;
    for (int f0 = 0; f0 < F; f0 += 256) {
        const int f = f0 + (int)threadIdx.x;
        const bool on = f < F && kp[f] != 0;
        const unsigned long long m = __ballot(on);
        const int before = __builtin_popcountll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = __builtin_popcountll(m);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; ++w) off += wsum[w];
        const int tot = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        if (on) {
            const int c = f / p.kmax, j = f - c * p.kmax;
            const size_t row = (size_t)g * F + off + before;
            const float* bb = bx + (size_t)(t0 + j) * 4;
            p.out_boxes[row * 4 + 0] = bb[0] / p.w; p.out_boxes[row * 4 + 1] = bb[1] / p.h;
            p.out_boxes[row * 4 + 2] = bb[2] / p.w; p.out_boxes[row * 4 + 3] = bb[3] / p.h;
            p.out_scores[row] = sc[(size_t)(t0 + j) * ss + c];
            p.out_cls[row] = c;
            p.out_tube[row] = j;
        }
        base += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) p.counts[g] = base;
}

}  // namespace step

using namespace step;

extern "C" {

int step_detect_compact(const uint8_t* keep, const float* const* boxes, const float* const* scores, const long long* score_strides,
                        const int32_t* tube_start, int I, int B, int NC, int kmax, float width, float height, float* out_boxes,
                        float* out_scores, long long* out_cls, long long* out_tube, int32_t* counts, step_stream_t stream) {
    if (I < 0 || I > STEP_DETECT_ITERS_MAX || B < 0 || NC < 0 || kmax < 0) return STEP_E_SHAPE;
    if (I == 0 || B == 0) return STEP_OK;
    if (!counts) return STEP_E_NULL;
    const bool none = NC == 0 || kmax == 0;                    // (no flags: the launch only writes the zero counts)
    if (!none && (!keep || !boxes || !scores || !score_strides || !tube_start || !out_boxes || !out_scores || !out_cls || !out_tube)) return STEP_E_NULL;
    DetectCompactParams p;
    for (int i = 0; i < STEP_DETECT_ITERS_MAX; ++i) {
        const bool in = i < I && !none;
        p.boxes[i] = in ? boxes[i] : nullptr; p.scores[i] = in ? scores[i] : nullptr; p.score_stride[i] = in ? score_strides[i] : 0;
        if (in && (!p.boxes[i] || !p.scores[i] || p.score_stride[i] < NC)) return STEP_E_SHAPE;
    }
    p.keep = keep; p.start = tube_start; p.B = B; p.NC = NC; p.kmax = kmax; p.w = width; p.h = height;
    p.out_boxes = out_boxes; p.out_scores = out_scores; p.out_cls = out_cls; p.out_tube = out_tube; p.counts = counts;
    STEP_LAUNCH((detect_compact_kernel), dim3((unsigned)(I * B)), dim3(256), stream, p);
    return STEP_LAUNCH_CHECK();
}

size_t step_nms_scratch_bytes(int G, int kmax) {
    if (G <= 0 || kmax <= 64) return 0;
    return (size_t)G * kmax * 5 + 16;
}

int step_nms_batched(const float* boxes, const float* scores, const int32_t* counts, int G, int kmax, float threshold,
                     uint8_t* keep, void* scratch, step_stream_t stream) {
    return nms_batched_t<float>(boxes, scores, counts, G, kmax, threshold, keep, scratch, stream);
}

int step_nms_batched_f64(const double* boxes, const double* scores, const int32_t* counts, int G, int kmax, float threshold,
                         uint8_t* keep, void* scratch, step_stream_t stream) {
    return nms_batched_t<double>(boxes, scores, counts, G, kmax, threshold, keep, scratch, stream);
}

int step_detect_nms(const float* prob, long long prob_stride, int NC, const float* loc, long long loc_stride, const int32_t* tube_start,
                    const int32_t* tube_count, int B, int kmax, float conf_thresh, float nms_thresh, float width, float height,
                    uint8_t* keep, float* boxes_out, step_stream_t stream) {
    if (B < 0 || NC < 0 || kmax < 0 || prob_stride < NC || loc_stride < 4) return STEP_E_SHAPE;
    if (kmax > 64) return STEP_E_UNSUPPORTED;                 // (more tubes per clip: mask + step_nms_batched, as before)
    if (B == 0 || NC == 0 || kmax == 0) return STEP_OK;
    if (!prob || !loc || !tube_start || !tube_count || !keep) return STEP_E_NULL;
    STEP_LAUNCH((detect_nms_wave_kernel), dim3((unsigned)(B * NC)), dim3(64), stream, prob, prob_stride, NC, loc, loc_stride, tube_start,
                tube_count, kmax, conf_thresh, nms_thresh, width, height, keep, boxes_out);
    return STEP_LAUNCH_CHECK();
}

}  // extern "C"
