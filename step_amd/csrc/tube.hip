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

}  // namespace step

using namespace step;

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
