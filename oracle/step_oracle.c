/*
 * step_oracle.c -- CPU restatement of the STEP ROI / NMS operators.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker.  The product path is the HIP
 * library built from step_amd/csrc (include/step_amd.h) and it never falls
 * back to this code.
 *
 * Every function restates one reference routine (paths below are relative
 * to /root/reference/external/maskrcnn_benchmark/csrc) in plain C, serial,
 * fp32, NCHW -- the reference's own layout -- in the reference's own order
 * of floating-point operations.  Build with -ffp-contract=off so that no
 * fused multiply-add is introduced (the reference CPU build is plain x86-64
 * SSE2 code without FMA).
 *
 * Pinning (see oracle/README.md, tests/test_oracle_vs_reference.py):
 *   orc_roi_align_forward, orc_nms  : checked bit-for-bit against the
 *       reference's own C++ (oracle/_ref/_C.so, compiled from the reference
 *       sources where they lie) and against tests/golden/roi_nms_golden.npz.
 *   orc_roi_align_backward          : the reference has no CPU implementation
 *       (ROIAlign.h:68); pinned indirectly as the exact adjoint of the
 *       forward (<fwd(x),g> == <x,bwd(g)>) -- tests/test_oracle_golden.py (test_roi_align_backward_is_adjoint).
 *   orc_roi_pool_forward/backward   : the reference has no CPU implementation
 *       (ROIPool.h:47,68), so no reference output can pin it: PARITY UNPINNED
 *       BY THE REFERENCE.  Cross-checked instead against an independent
 *       implementation of the same published operator (torch's
 *       adaptive_max_pool2d on the cropped RoI: values and argmax positions,
 *       in-bounds integer RoIs) and, for the backward, as the exact scatter of
 *       the argmax -- tests/test_oracle_golden.py
 *       (test_roi_pool_restatement_against_an_independent_implementation).
 */
#include <math.h>
#include <float.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* One bilinear sample: tap positions and weights.
 * cpu/ROIAlign_cpu.cpp:41-135 (pre_calc_for_bilinear_interpolate) and
 * cuda/ROIAlign_cuda.cu:39-86,149-199 compute the same quantities. */
typedef struct {
    int y_low, y_high, x_low, x_high; /* -1 everywhere when the sample is void */
    float w1, w2, w3, w4;
} orc_tap_t;

static orc_tap_t orc_bilinear_tap(int height, int width, float y, float x)
{
    orc_tap_t t;
    /* ROIAlign_cpu.cpp:73-86: samples outside [-1,H] x [-1,W] contribute 0 */
    if (y < -1.0 || y > height || x < -1.0 || x > width) {
        t.y_low = t.y_high = t.x_low = t.x_high = -1;
        t.w1 = t.w2 = t.w3 = t.w4 = 0.f;
        return t;
    }
    /* ROIAlign_cpu.cpp:88-93 */
    if (y <= 0) y = 0;
    if (x <= 0) x = 0;
    /* ROIAlign_cpu.cpp:95-112 */
    int y_low = (int)y;
    int x_low = (int)x;
    int y_high, x_high;
    if (y_low >= height - 1) {
        y_high = y_low = height - 1;
        y = (float)y_low;
    } else {
        y_high = y_low + 1;
    }
    if (x_low >= width - 1) {
        x_high = x_low = width - 1;
        x = (float)x_low;
    } else {
        x_high = x_low + 1;
    }
    /* ROIAlign_cpu.cpp:114-117 (1. - l is exact in double, so the float
     * result equals the correctly rounded float subtraction) */
    float ly = y - y_low;
    float lx = x - x_low;
    float hy = (float)(1. - ly), hx = (float)(1. - lx);
    t.w1 = hy * hx; t.w2 = hy * lx; t.w3 = ly * hx; t.w4 = ly * lx;
    t.y_low = y_low; t.y_high = y_high; t.x_low = x_low; t.x_high = x_high;
    return t;
}

/* Per-ROI geometry shared by forward and backward.
 * cpu/ROIAlign_cpu.cpp:166-192, cuda/ROIAlign_cuda.cu:101-124. */
typedef struct {
    int batch;
    float start_w, start_h, bin_h, bin_w;
    int grid_h, grid_w;
    float count;
} orc_roi_geom_t;

static orc_roi_geom_t orc_roi_geom(const float* roi, float scale, int ph, int pw, int sampling_ratio)
{
    orc_roi_geom_t g;
    g.batch = (int)roi[0];               /* float batch index truncated, :162 */
    g.start_w = roi[1] * scale;          /* no rounding, :166-169 */
    g.start_h = roi[2] * scale;
    float end_w = roi[3] * scale;
    float end_h = roi[4] * scale;
    float roi_w = fmaxf(end_w - g.start_w, 1.f);  /* malformed ROIs -> 1x1, :176-177 */
    float roi_h = fmaxf(end_h - g.start_h, 1.f);
    g.bin_h = roi_h / (float)ph;         /* :178-179 */
    g.bin_w = roi_w / (float)pw;
    g.grid_h = (sampling_ratio > 0) ? sampling_ratio : (int)ceilf(roi_h / ph); /* :182-186 */
    g.grid_w = (sampling_ratio > 0) ? sampling_ratio : (int)ceilf(roi_w / pw);
    g.count = (float)(g.grid_h * g.grid_w);   /* :189 */
    return g;
}

/* ROIAlign forward, NCHW.  cpu/ROIAlign_cpu.cpp:137-243.
 * in [B,C,H,W], rois [K,5] (batch,x1,y1,x2,y2), out [K,C,ph,pw]. */
ORC_API void orc_roi_align_forward(const float* in, const float* rois, int K, int C, int H, int W,
                                   int ph, int pw, float scale, int sampling_ratio, float* out)
{
    for (int n = 0; n < K; n++) {
        orc_roi_geom_t g = orc_roi_geom(rois + 5 * n, scale, ph, pw, sampling_ratio);
        int ntap = ph * pw * g.grid_h * g.grid_w;
        orc_tap_t* taps = (orc_tap_t*)malloc(sizeof(orc_tap_t) * (size_t)(ntap > 0 ? ntap : 1));
        int ti = 0;
        for (int p = 0; p < ph; p++)
            for (int q = 0; q < pw; q++)
                for (int iy = 0; iy < g.grid_h; iy++) {
                    /* ROIAlign_cpu.cpp:61-63 */
                    const float yy = g.start_h + p * g.bin_h +
                                     (float)(iy + .5f) * g.bin_h / (float)g.grid_h;
                    for (int ix = 0; ix < g.grid_w; ix++) {
                        const float xx = g.start_w + q * g.bin_w +
                                         (float)(ix + .5f) * g.bin_w / (float)g.grid_w;
                        taps[ti++] = orc_bilinear_tap(H, W, yy, xx);
                    }
                }
        for (int c = 0; c < C; c++) {
            const float* d = in + ((size_t)g.batch * C + c) * H * W;
            float* o = out + ((size_t)n * C + c) * ph * pw;
            ti = 0;
            for (int p = 0; p < ph; p++)
                for (int q = 0; q < pw; q++) {
                    float v = 0.f;
                    for (int s = 0; s < g.grid_h * g.grid_w; s++) {
                        orc_tap_t t = taps[ti++];
                        if (t.y_low < 0) { v += 0.f; continue; } /* void sample: pos=0,w=0, :75-86 */
                        /* ROIAlign_cpu.cpp:221-224: left-to-right sum of the 4 taps */
                        v += t.w1 * d[t.y_low * W + t.x_low] + t.w2 * d[t.y_low * W + t.x_high] +
                             t.w3 * d[t.y_high * W + t.x_low] + t.w4 * d[t.y_high * W + t.x_high];
                    }
                    v /= g.count;          /* :229 */
                    o[p * pw + q] = v;
                }
        }
        free(taps);
    }
}

/* ROIAlign backward, NCHW.  cuda/ROIAlign_cuda.cu:201-278 executed serially
 * in index order (the reference uses atomicAdd, so its own summation order is
 * unspecified).  gin [B,C,H,W] is zeroed here (ROIAlign_cuda.cu:340). */
ORC_API void orc_roi_align_backward(const float* grad, const float* rois, int K, int B, int C, int H,
                                    int W, int ph, int pw, float scale, int sampling_ratio, float* gin)
{
    memset(gin, 0, sizeof(float) * (size_t)B * C * H * W);
    for (int n = 0; n < K; n++) {
        orc_roi_geom_t g = orc_roi_geom(rois + 5 * n, scale, ph, pw, sampling_ratio);
        for (int c = 0; c < C; c++) {
            float* d = gin + ((size_t)g.batch * C + c) * H * W;
            const float* go = grad + ((size_t)n * C + c) * ph * pw;
            for (int p = 0; p < ph; p++)
                for (int q = 0; q < pw; q++) {
                    const float gtop = go[p * pw + q];
                    for (int iy = 0; iy < g.grid_h; iy++) {
                        const float y = g.start_h + p * g.bin_h +
                                        (float)(iy + .5f) * g.bin_h / (float)g.grid_h;
                        for (int ix = 0; ix < g.grid_w; ix++) {
                            const float x = g.start_w + q * g.bin_w +
                                            (float)(ix + .5f) * g.bin_w / (float)g.grid_w;
                            orc_tap_t t = orc_bilinear_tap(H, W, y, x);
                            /* ROIAlign_cuda.cu:262-273 */
                            float g1 = gtop * t.w1 / g.count;
                            float g2 = gtop * t.w2 / g.count;
                            float g3 = gtop * t.w3 / g.count;
                            float g4 = gtop * t.w4 / g.count;
                            if (t.x_low >= 0 && t.x_high >= 0 && t.y_low >= 0 && t.y_high >= 0) {
                                d[t.y_low * W + t.x_low] += g1;
                                d[t.y_low * W + t.x_high] += g2;
                                d[t.y_high * W + t.x_low] += g3;
                                d[t.y_high * W + t.x_high] += g4;
                            }
                        }
                    }
                }
        }
    }
}

static int orc_imin(int a, int b) { return a < b ? a : b; }
static int orc_imax(int a, int b) { return a > b ? a : b; }

/* ROIPool forward, NCHW.  cuda/ROIPool_cuda.cu:40-101.  PARITY UNPINNED
 * (no reference CPU implementation exists). */
ORC_API void orc_roi_pool_forward(const float* in, const float* rois, int K, int C, int H, int W,
                                  int ph, int pw, float scale, float* out, int32_t* argmax)
{
    for (int n = 0; n < K; n++) {
        const float* r = rois + 5 * n;
        int batch = (int)r[0];
        /* ROIPool_cuda.cu:53-56: round() = half away from zero */
        int sw = (int)roundf(r[1] * scale);
        int sh = (int)roundf(r[2] * scale);
        int ew = (int)roundf(r[3] * scale);
        int eh = (int)roundf(r[4] * scale);
        int rw = orc_imax(ew - sw + 1, 1);   /* :59-60 */
        int rh = orc_imax(eh - sh + 1, 1);
        float bin_h = (float)rh / (float)ph; /* :61-64 */
        float bin_w = (float)rw / (float)pw;
        for (int c = 0; c < C; c++) {
            const float* d = in + ((size_t)batch * C + c) * H * W;
            for (int p = 0; p < ph; p++)
                for (int q = 0; q < pw; q++) {
                    /* :66-73 */
                    int hstart = (int)floorf((float)p * bin_h);
                    int wstart = (int)floorf((float)q * bin_w);
                    int hend = (int)ceilf((float)(p + 1) * bin_h);
                    int wend = (int)ceilf((float)(q + 1) * bin_w);
                    /* :76-79 */
                    hstart = orc_imin(orc_imax(hstart + sh, 0), H);
                    hend = orc_imin(orc_imax(hend + sh, 0), H);
                    wstart = orc_imin(orc_imax(wstart + sw, 0), W);
                    wend = orc_imin(orc_imax(wend + sw, 0), W);
                    int empty = (hend <= hstart) || (wend <= wstart);
                    float maxval = empty ? 0.f : -FLT_MAX;  /* :83 */
                    int maxidx = -1;                          /* :85 */
                    for (int h = hstart; h < hend; ++h)
                        for (int w = wstart; w < wend; ++w) {
                            int bi = h * W + w;
                            if (d[bi] > maxval) { maxval = d[bi]; maxidx = bi; }
                        }
                    size_t oi = (((size_t)n * C + c) * ph + p) * pw + q;
                    out[oi] = maxval;
                    argmax[oi] = maxidx;
                }
        }
    }
}

/* ROIPool backward, NCHW.  cuda/ROIPool_cuda.cu:103-132 (serial order). */
ORC_API void orc_roi_pool_backward(const float* grad, const int32_t* argmax, const float* rois, int K,
                                   int B, int C, int H, int W, int ph, int pw, float* gin)
{
    memset(gin, 0, sizeof(float) * (size_t)B * C * H * W);
    for (int n = 0; n < K; n++) {
        int batch = (int)rois[5 * n];
        for (int c = 0; c < C; c++) {
            float* d = gin + ((size_t)batch * C + c) * H * W;
            for (int i = 0; i < ph * pw; i++) {
                size_t oi = ((size_t)n * C + c) * ph * pw + i;
                int a = argmax[oi];
                if (a != -1) d[a] += grad[oi];
            }
        }
    }
}

/* Stable descending argsort of scores (ties: lower original index first).
 * The reference uses scores.sort(0, descending=true) (cpu/nms_cpu.cpp:48),
 * whose tie order is unspecified; the oracle pins "lower index first". */
typedef struct { float s; int64_t i; } orc_sk_t;
static int orc_sk_cmp(const void* a, const void* b)
{
    const orc_sk_t* x = (const orc_sk_t*)a; const orc_sk_t* y = (const orc_sk_t*)b;
    const int nx = x->s != x->s, ny = y->s != y->s;   /* torch's sort puts NaN first in descending order (observed on _ref/_C.so) */
    if (nx != ny) return nx ? -1 : 1;
    if (x->s > y->s) return -1;
    if (x->s < y->s) return 1;
    return (x->i < y->i) ? -1 : (x->i > y->i);
}

/* Greedy NMS.  cpu/nms_cpu.cpp:29-89.  boxes [n,4] (x1,y1,x2,y2), scores [n].
 * Writes the kept ORIGINAL indices in ascending order to keep[] (:88) and
 * returns their number. */
ORC_API int64_t orc_nms(const float* boxes, const float* scores, int64_t n, float threshold, int64_t* keep)
{
    if (n <= 0) return 0;                              /* :37-39 */
    orc_sk_t* ord = (orc_sk_t*)malloc(sizeof(orc_sk_t) * (size_t)n);
    float* area = (float*)malloc(sizeof(float) * (size_t)n);
    uint8_t* sup = (uint8_t*)calloc((size_t)n, 1);
    for (int64_t i = 0; i < n; i++) {
        const float* b = boxes + 4 * i;
        float w = b[2] - b[0]; w = w + 1.f;            /* :46: (x2-x1+1)*(y2-y1+1), each op rounded */
        float h = b[3] - b[1]; h = h + 1.f;
        area[i] = w * h;
        ord[i].s = scores[i]; ord[i].i = i;
    }
    qsort(ord, (size_t)n, sizeof(orc_sk_t), orc_sk_cmp);
    for (int64_t _i = 0; _i < n; _i++) {              /* :62-86 */
        int64_t i = ord[_i].i;
        if (sup[i]) continue;
        const float* bi = boxes + 4 * i;
        for (int64_t _j = _i + 1; _j < n; _j++) {
            int64_t j = ord[_j].i;
            if (sup[j]) continue;
            const float* bj = boxes + 4 * j;
            float xx1 = fmaxf(bi[0], bj[0]);
            float yy1 = fmaxf(bi[1], bj[1]);
            float xx2 = fminf(bi[2], bj[2]);
            float yy2 = fminf(bi[3], bj[3]);
            float w = xx2 - xx1; w = w + 1.f; w = fmaxf(0.f, w);
            float h = yy2 - yy1; h = h + 1.f; h = fmaxf(0.f, h);
            float inter = w * h;
            float den = area[i] + area[j]; den = den - inter;
            float ovr = inter / den;
            if (ovr >= threshold) sup[j] = 1;          /* :84: >= (the CUDA op uses >) */
        }
    }
    int64_t m = 0;
    for (int64_t i = 0; i < n; i++) if (!sup[i]) keep[m++] = i;  /* :88 nonzero(suppressed==0) */
    free(ord); free(area); free(sup);
    return m;
}

/* The same in double precision: the reference dispatches AT_DISPATCH_FLOATING_TYPES (cpu/nms_cpu.cpp:95), so fp64 boxes / scores are
 * compared in fp64 (the threshold stays the float argument of nms_cpu_kernel, promoted in `ovr >= threshold`). */
typedef struct { double s; int64_t i; } orc_skd_t;
static int orc_skd_cmp(const void* a, const void* b)
{
    const orc_skd_t* x = (const orc_skd_t*)a; const orc_skd_t* y = (const orc_skd_t*)b;
    const int nx = x->s != x->s, ny = y->s != y->s;
    if (nx != ny) return nx ? -1 : 1;
    if (x->s > y->s) return -1;
    if (x->s < y->s) return 1;
    return (x->i < y->i) ? -1 : (x->i > y->i);
}
ORC_API int64_t orc_nms_f64(const double* boxes, const double* scores, int64_t n, float threshold, int64_t* keep)
{
    if (n <= 0) return 0;
    orc_skd_t* ord = (orc_skd_t*)malloc(sizeof(orc_skd_t) * (size_t)n);
    double* area = (double*)malloc(sizeof(double) * (size_t)n);
    uint8_t* sup = (uint8_t*)calloc((size_t)n, 1);
    for (int64_t i = 0; i < n; i++) {
        const double* b = boxes + 4 * i;
        double w = b[2] - b[0]; w = w + 1.0;
        double h = b[3] - b[1]; h = h + 1.0;
        area[i] = w * h;
        ord[i].s = scores[i]; ord[i].i = i;
    }
    qsort(ord, (size_t)n, sizeof(orc_skd_t), orc_skd_cmp);
    for (int64_t _i = 0; _i < n; _i++) {
        int64_t i = ord[_i].i;
        if (sup[i]) continue;
        const double* bi = boxes + 4 * i;
        for (int64_t _j = _i + 1; _j < n; _j++) {
            int64_t j = ord[_j].i;
            if (sup[j]) continue;
            const double* bj = boxes + 4 * j;
            double xx1 = fmax(bi[0], bj[0]);
            double yy1 = fmax(bi[1], bj[1]);
            double xx2 = fmin(bi[2], bj[2]);
            double yy2 = fmin(bi[3], bj[3]);
            double w = xx2 - xx1; w = w + 1.0; w = fmax(0.0, w);
            double h = yy2 - yy1; h = h + 1.0; h = fmax(0.0, h);
            double inter = w * h;
            double den = area[i] + area[j]; den = den - inter;
            double ovr = inter / den;
            if (ovr >= threshold) sup[j] = 1;
        }
    }
    int64_t m = 0;
    for (int64_t i = 0; i < n; i++) if (!sup[i]) keep[m++] = i;
    free(ord); free(area); free(sup);
    return m;
}

/* Batched form used to check the tube-batched HIP nms: G independent groups,
 * group g has counts[g] boxes stored at boxes[g*kmax*4 ...], scores[g*kmax ...].
 * keep_mask[g*kmax + i] = 1 if box i of group g survives. */
ORC_API void orc_nms_batched(const float* boxes, const float* scores, const int32_t* counts, int G,
                             int kmax, float threshold, uint8_t* keep_mask)
{
    int64_t* tmp = (int64_t*)malloc(sizeof(int64_t) * (size_t)(kmax > 0 ? kmax : 1));
    memset(keep_mask, 0, (size_t)G * kmax);
    for (int g = 0; g < G; g++) {
        int64_t m = orc_nms(boxes + (size_t)g * kmax * 4, scores + (size_t)g * kmax, counts[g], threshold, tmp);
        for (int64_t i = 0; i < m; i++) keep_mask[(size_t)g * kmax + tmp[i]] = 1;
    }
    free(tmp);
}
