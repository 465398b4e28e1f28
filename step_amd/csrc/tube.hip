// step_amd/csrc/tube.hip -- the per-step tube bookkeeping of the multi-step inference driver as ONE launch.
//
// Replaces the host glue of utils/utils.py:68-129 (decode_coef x3 -> torch.cat -> valid_tubes -> flatten_tubes with the
// frame-index column; tube_utils.py:59-92,178-189,214-246), which the reference runs on the CPU through numpy per clip and
// which a tensor-op restatement turns into ~60 tiny element-wise launches per refinement step.  Pure fp32 element-wise
// arithmetic in the reference's operation order (no FMA contraction), one thread per (tube, output frame).
#include "common.h"

#pragma clang fp contract(off)

namespace step {

struct TubeParams {
    const float* tubes; const float* local_loc; const float* first_loc; const float* last_loc; const int32_t* clip_of;
    float* pred_loc; float* pred_first; float* pred_last; float* next_tubes;
    int N, T, Tw, first_off, last_off, extend;
    float width, height;
};

__device__ __forceinline__ void decode_box(const float* a /*x1,y1,x2,y2*/, const float* d, float (&o)[4]) {
    const float w = a[2] - a[0] + 1.0f, h = a[3] - a[1] + 1.0f;                 // get_center_size (tube_utils.py:127-134)
    const float x = a[0] + 0.5f * w, y = a[1] + 0.5f * h;
    const float px = w * d[0] + x, py = h * d[1] + y;                             // decode_coef (tube_utils.py:178-189)
    const float pw = w * expf(d[2]), ph = h * expf(d[3]);
    o[0] = px - 0.5f * pw; o[1] = py - 0.5f * ph; o[2] = px + 0.5f * pw - 1.0f; o[3] = py + 0.5f * ph - 1.0f;
}

__global__ void tube_update_kernel(TubeParams p) {
    const int Tn = p.extend ? p.T + 2 * p.Tw : p.T;
    const int Tall = p.T + 2 * p.Tw;                          // every tube decodes first | local | last (history wants all three)
    const long long total = (long long)p.N * Tall;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)blockDim.x * gridDim.x) {
        const int n = (int)(idx / Tall), u = (int)(idx % Tall);
        const float* anc; const float* del; float* dst; int tn;
        if (u < p.Tw) {                                       // first-frame neighbours: anchors = the tube's first chunk
            anc = p.tubes + ((size_t)n * p.T + p.first_off + u) * 5 + 1;
            del = p.first_loc + ((size_t)n * p.Tw + u) * 4;
            dst = p.pred_first + ((size_t)n * p.Tw + u) * 4;
            tn = p.extend ? u : -1;
        } else if (u < p.Tw + p.T) {
            const int t = u - p.Tw;
            anc = p.tubes + ((size_t)n * p.T + t) * 5 + 1;
            del = p.local_loc + ((size_t)n * p.T + t) * 4;
            dst = p.pred_loc + ((size_t)n * p.T + t) * 4;
            tn = p.extend ? u : t;
        } else {
            const int t = u - p.Tw - p.T;
            anc = p.tubes + ((size_t)n * p.T + p.last_off + t) * 5 + 1;
            del = p.last_loc + ((size_t)n * p.Tw + t) * 4;
            dst = p.pred_last + ((size_t)n * p.Tw + t) * 4;
            tn = p.extend ? u : -1;
        }
        float o[4];
        decode_box(anc, del, o);
        dst[0] = o[0]; dst[1] = o[1]; dst[2] = o[2]; dst[3] = o[3];
        if (tn >= 0) {                                        // the next step's proposal: valid_tubes + the frame-index column
            float x1 = o[0] < 0.0f ? 0.0f : o[0], y1 = o[1] < 0.0f ? 0.0f : o[1];          // clamp(min = 0): NaN stays NaN
            float x2 = o[2] > p.width ? p.width : o[2], y2 = o[3] > p.height ? p.height : o[3];
            if (!((x1 < x2 - 2.0f) && (y1 < y2 - 2.0f))) { x1 = 0.0f; y1 = 0.0f; x2 = p.width; y2 = p.height; }
            float* q = p.next_tubes + ((size_t)n * Tn + tn) * 5;
            q[0] = (float)p.clip_of[n] * (float)Tn + (float)tn;
            q[1] = x1; q[2] = y1; q[3] = x2; q[4] = y2;
        }
    }
}

// ---- training sample selection, device front end -------------------------------------------------------------
// What utils/utils.py:179-214 (train_select) and utils/tube_utils.py:269-351 compute per clip on the host from a previous step's
// predictions, for EVERY refined tube in one launch: the class scores averaged over the tube's frames, the three predicted tubes
// through valid_tubes, and the IoU of the tube's middle-frame box with each of its clip's ground-truth boxes.  fp32 arithmetic in
// the reference's (numpy's) operation order: the mean is the sequential sum over frames divided by T; box_iou without the +1
// convention, zero unless both overlap extents are positive; a pair with an all-zero (padding) tube gives 0.
struct SelectParams {
    const float* prob; const float* loc; const float* first; const float* last; const int32_t* clip_of; const float* gt; const int32_t* gt_count;
    float* mean_prob; float* vloc; float* vfirst; float* vlast; float* iou;
    int N, T, Tw, NC, Gmax;
    float width, height;
};

__device__ __forceinline__ void valid_box(const float* b, float w, float h, float* o) {        // tube_utils.py:59-92
    float x1 = fmaxf(0.0f, b[0]), y1 = fmaxf(0.0f, b[1]), x2 = fminf(w, b[2]), y2 = fminf(h, b[3]);
    // (np.maximum / np.minimum propagate NaN where fmaxf / fminf drop it: a NaN coordinate fails the reference's `<` test)
    const bool nan_in = (b[0] != b[0]) || (b[1] != b[1]) || (b[2] != b[2]) || (b[3] != b[3]);
    if (nan_in || !((x1 < x2 - 2.0f) && (y1 < y2 - 2.0f))) { x1 = 0.0f; y1 = 0.0f; x2 = w; y2 = h; }
    o[0] = x1; o[1] = y1; o[2] = x2; o[3] = y2;
}

__global__ void select_prepare_kernel(SelectParams p) {
    const int n = blockIdx.x;
    for (int c = threadIdx.x; c < p.NC; c += blockDim.x) {
        float s = p.prob[((size_t)n * p.T) * p.NC + c];
        for (int t = 1; t < p.T; ++t) s = s + p.prob[((size_t)n * p.T + t) * p.NC + c];
        p.mean_prob[(size_t)n * p.NC + c] = s / (float)p.T;
    }
    for (int t = threadIdx.x; t < p.T; t += blockDim.x) valid_box(p.loc + ((size_t)n * p.T + t) * 4, p.width, p.height, p.vloc + ((size_t)n * p.T + t) * 4);
    if (p.first)
        for (int t = threadIdx.x; t < p.Tw; t += blockDim.x) {
            valid_box(p.first + ((size_t)n * p.Tw + t) * 4, p.width, p.height, p.vfirst + ((size_t)n * p.Tw + t) * 4);
            valid_box(p.last + ((size_t)n * p.Tw + t) * 4, p.width, p.height, p.vlast + ((size_t)n * p.Tw + t) * 4);
        }
    const int b = p.clip_of[n];
    for (int g = threadIdx.x; g < p.Gmax; g += blockDim.x) {
        float v = 0.0f;
        if (g < p.gt_count[b]) {
            float a[4];
            valid_box(p.loc + ((size_t)n * p.T + p.T / 2) * 4, p.width, p.height, a);        // the candidate's middle frame (after valid_tubes)
            const float* q = p.gt + ((size_t)b * p.Gmax + g) * 4;
            const bool live = (((q[0] + q[1]) + q[2]) + q[3]) != 0.0f && (((a[0] + a[1]) + a[2]) + a[3]) != 0.0f;     // bool(np.sum(tube))
            if (live) {
                const float iw = fmaxf(fminf(q[2], a[2]) - fmaxf(q[0], a[0]), 0.0f);
                const float ih = fmaxf(fminf(q[3], a[3]) - fmaxf(q[1], a[1]), 0.0f);
                const float inter = (iw > 0.0f && ih > 0.0f) ? iw * ih : 0.0f;
                const float uni = (q[2] - q[0]) * (q[3] - q[1]) + (a[2] - a[0]) * (a[3] - a[1]) - inter;
                v = inter / uni;
            }
        }
        p.iou[(size_t)n * p.Gmax + g] = v;
    }
}

}  // namespace step

using namespace step;

extern "C" int step_select_prepare(const float* prob, const float* loc, const float* first, const float* last, int N, int T, int Tw, int NC,
                                   const int32_t* clip_of, const float* gt_mid, const int32_t* gt_count, int Gmax, float width, float height,
                                   float* mean_prob, float* vloc, float* vfirst, float* vlast, float* iou, step_stream_t stream) {
    if (N < 0 || T <= 0 || NC <= 0 || Gmax < 0 || Tw < 0) return STEP_E_SHAPE;
    if (N == 0) return STEP_OK;
    if (!prob || !loc || !clip_of || !mean_prob || !vloc || (Gmax && (!gt_mid || !gt_count || !iou))) return STEP_E_NULL;
    if ((first != nullptr) != (last != nullptr) || (first && (!vfirst || !vlast || Tw <= 0))) return STEP_E_NULL;
    SelectParams p;
    p.prob = prob; p.loc = loc; p.first = first; p.last = last; p.clip_of = clip_of; p.gt = gt_mid; p.gt_count = gt_count;
    p.mean_prob = mean_prob; p.vloc = vloc; p.vfirst = vfirst; p.vlast = vlast; p.iou = iou;
    p.N = N; p.T = T; p.Tw = Tw; p.NC = NC; p.Gmax = Gmax; p.width = width; p.height = height;
    STEP_LAUNCH(select_prepare_kernel, dim3((unsigned)N), dim3(64), stream, p);
    return STEP_LAUNCH_CHECK();
}

extern "C" int step_tube_update(const float* tubes, int N, int T, const float* local_loc, const float* first_loc, const float* last_loc,
                                int Tw, int first_off, int last_off, const int32_t* clip_of, int extend, float width, float height,
                                float* pred_loc, float* pred_first, float* pred_last, float* next_tubes, step_stream_t stream) {
    if (N < 0 || T <= 0 || Tw <= 0 || first_off < 0 || last_off < 0 || first_off + Tw > T || last_off + Tw > T) return STEP_E_SHAPE;
    if (N == 0) return STEP_OK;
    if (!tubes || !local_loc || !first_loc || !last_loc || !clip_of || !pred_loc || !pred_first || !pred_last || !next_tubes) return STEP_E_NULL;
    TubeParams p;
    p.tubes = tubes; p.local_loc = local_loc; p.first_loc = first_loc; p.last_loc = last_loc; p.clip_of = clip_of;
    p.pred_loc = pred_loc; p.pred_first = pred_first; p.pred_last = pred_last; p.next_tubes = next_tubes;
    p.N = N; p.T = T; p.Tw = Tw; p.first_off = first_off; p.last_off = last_off; p.extend = extend ? 1 : 0;
    p.width = width; p.height = height;
    const long long total = (long long)N * (T + 2 * Tw);
    long long g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    STEP_LAUNCH(tube_update_kernel, dim3((unsigned)g), dim3(256), stream, p);
    return STEP_LAUNCH_CHECK();
}
