/*
 * include/step_amd.h -- C ABI of libstep_amd.so, the MI355X (gfx950) implementation of the STEP
 * hot path: ROIAlign / ROIPool / NMS operators and the I3D conv / pool building blocks.
 *
 * This is the drop-in boundary.  It replaces the five pybind11 symbols of the reference's native
 * extension `external.maskrcnn_benchmark.roi_layers._C`
 *     (/root/reference/external/maskrcnn_benchmark/csrc/vision.cpp:30-36)
 * and the cuDNN calls behind torch.nn.Conv3d/BatchNorm3d/MaxPool3d/AvgPool3d/Conv2d/Linear that
 * models/i3dpt.py and models/two_branch.py lean on -- forward, and for the training step (train.py:257-348) their
 * backward (data / weight gradients, pool and activation gradients) and the Adam update.  INTEGRATION.md shows the
 * binding a maintainer of the reference adds.
 *
 * Conventions
 *   - plain C: raw DEVICE pointers, explicit sizes, a hipStream_t passed as void*.  No torch types.
 *   - the library never allocates, frees or retains memory: outputs and scratch are the caller's
 *     (reference: outputs are fresh ATen tensors, ROIAlign_cuda.cu:295,340; scratch THCudaMalloc,
 *     nms.cu:113).
 *   - every entry point is asynchronous on `stream`, re-entrant and stateless (nn.DataParallel
 *     calls replicas from one thread per GPU; the caller selects the device).  The library reads no environment
 *     variable; its only process-wide state is the table of planner options below (step_set_option).
 *   - return value: 0 ok; <0 bad argument (STEP_E_*); >0 a hipError_t from the launch.
 *     (reference: AT_ASSERTM / THCudaCheck throw -> Python RuntimeError; our Python layer raises
 *     RuntimeError on any non-zero status.)
 *   - empty inputs (K == 0, n == 0) return 0 without launching (ROIAlign_cuda.cu:302-305,
 *     nms.h:41-42).
 */
#ifndef STEP_AMD_H
#define STEP_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STEP_API __attribute__((visibility("default")))

typedef void* step_stream_t; /* hipStream_t */

/* element types of feature / activation tensors */
enum { STEP_F32 = 0, STEP_BF16 = 1, STEP_F16 = 2 };
/* physical layout of a 4-D feature map passed to the ROI operators */
enum { STEP_NCHW = 0, STEP_NHWC = 1 };

enum {
    STEP_OK = 0,
    STEP_E_DTYPE = -1,
    STEP_E_SHAPE = -2,
    STEP_E_NULL = -3,
    STEP_E_UNSUPPORTED = -4,
    STEP_E_ALIGN = -5
};

/* Library identity: returns "step_amd <version> gfx950"; abi is bumped on any signature change. */
STEP_API const char* step_version(void);
STEP_API int step_abi_version(void);

/* Diagnostic (bench.py): `workgroups` x 256 threads each issue `iters` x 4 back-to-back v_mfma_f32_32x32x16_bf16 per wavefront on
 * pseudo-random bf16 operands (the clock depends on how many operand bits toggle: near-constant operands run ~0.45 GHz higher) and
 * nothing else; out[3 * wg + {0, 1, 2}] = shader cycles (s_memtime), 100 MHz ticks (s_memrealtime) of that loop, and a nonzero
 * flag.  One workgroup per CU gives the clock and the dense 16-bit matrix rate the box SUSTAINS -- MI355X is power-managed: the
 * boxes this was developed on settle at ~1.9 GHz = ~2.0 PFLOP/s, not the 2.4 GHz / 2.5 PFLOP/s of the datasheet roofline. */
STEP_API int step_mfma_clock_probe(unsigned long long* out, int workgroups, int iters, step_stream_t stream);

/* Diagnostic (bench.py): the EFFECTIVE shader clock while other work runs.  `workgroups` single-wavefront workgroups each read the shader
 * cycle counter and the 100 MHz real-time counter, sleep (s_sleep, no memory traffic) until `ticks_100mhz` ticks have passed and read both
 * again: out[2 * wg + {0, 1}] = shader cycles, 100 MHz ticks; cycles / (ticks * 10) = GHz.  Launched on a side stream beside a loop it
 * takes one wave slot; the driver's DPM state (sysfs / amd-smi) is NOT this number. */
STEP_API int step_clock_sample(unsigned long long* out, int workgroups, int ticks_100mhz, step_stream_t stream);

/* The other measured ceiling: a plain streaming copy of `bytes` (multiple of 16; src, dst 16-byte aligned, not overlapping) by
 * `workgroups` 256-thread workgroups, 16 B per lane, grid-stride.  2 * bytes / its duration on buffers well beyond the 256 MB
 * last-level cache is the HBM rate one launch reaches on this box -- what the pools / pointwise convs / ROIAlign are read against
 * (the datasheet's 8 TB/s is not reachable by any copy). */
STEP_API int step_hbm_stream_probe(const void* src, void* dst, size_t bytes, int workgroups, step_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Planner options.  The library reads NO environment variable: the launch planners' few tuning / test knobs are explicit
 * integers set through this entry point (process-wide, relaxed atomics: safe to call from any thread; a launch sees either
 * the old or the new value).  Every option has a default under which the planner decides by itself (its measured rules);
 * no option changes WHAT an entry point computes beyond fp32 summation order where the option's line says so -- they exist
 * so that tests can reach every kernel form at interpreter sizes and tools/ab_bench.py can time one form against another
 * in one process.  step_set_option returns STEP_E_SHAPE for an unknown id or a value outside the option's range.
 */
enum {
    STEP_OPT_CONV_IMPL = 0,    /* -1 auto | 0 tiled 4-wave kernel | 1 conv_tap, one tap per step | 2 conv_tap, two taps | 5 streaming pointwise GEMM for every 1x1x1 */
    STEP_OPT_CONV_NB,          /*  0 auto | 1..3 accumulator depth of conv_tap / conv_pw */
    STEP_OPT_CONV_WAVES,       /*  0 auto | 4 | 8 wavefronts per workgroup of conv_tap / conv_pw */
    STEP_OPT_CONV_PHASED,      /*  2 (default) two-phase conv_tap for 16-bit 3x3x3 and, at NB <= 2 on general boxes, 1x3x3 windows | 1: 3x3x3 only | 0 classic pipeline
                                    (bit-identical results) */
    STEP_OPT_CONV_GEN,         /*  93 (default): a general tile box when it needs <= this percent of the best power-of-two tiling's tiles; 0 never */
    STEP_OPT_CONV_GMODE,       /*  1 (default) conflict-free pixel assignment inside general boxes | 0 linear walk (bit-identical) */
    STEP_OPT_CONV_PWS,         /* -1 auto | 0 never | 1 wherever its contract allows: the weight-stationary pointwise stream (bit-identical) */
    STEP_OPT_CONV_SPLITK,      /*  1 (default) | 0: split-K form of few-row / deep-K pointwise layers (fixed-order sum either way) */
    STEP_OPT_CONV_TAIL,        /*  1 (default) | 0: partial last round of a one-channel-group conv_tap launch at NB = 1 (bit-identical) */
    STEP_OPT_CONV_SLOTS,       /*  0 (default: resident workgroups of the chip) | n: pretend the chip holds n workgroups (tests: the tail split at small sizes) */
    STEP_OPT_POOL_DIRECT,      /*  0 (default) | 1: every max pool on the general 27-tap kernel (tests) */
    STEP_OPT_WGRAD_MINPIX,     /*  0 (default: 512 for the atomics forms, 64 (fp32 MFMA) / 256 (16-bit) with a workspace) | n: least pixels per wavefront job of the per-tap weight gradient (fp32 summation order) */
    STEP_OPT_WGRAD16_LDS,      /*  1 (default): 3x3 windows on the LDS-tiled GEMM, pointwise layers on the pixel stream | 2: pointwise layers on the forms of early round 4 (LDS tiles / per tap) | 0: everything on the per-tap kernel (fp32 summation order differs between the three) */
    STEP_OPT_CONV_GROUP_PW,    /*  2^20 (default: always) | n: step_conv_forward_group carries a pointwise item inside the 3x3x3 members' grid when they are at most n workgroups; 0 never (bit-identical) */
    STEP_OPT_CLIP_VEC,         /*  1 (default): step_clip_from_u8 converts 16 pixels per thread with 16-byte accesses where H*W % 16 == 0 | 0: the per-pixel kernel (bit-identical) */
    STEP_OPT_CONV_NB_RULE,     /*  0 (default): conv_tap's accumulator depth minimises ROUNDS of the chip (the latency of one launch) | 1: minimises workgroups x per-workgroup time (the chip time of the launch: what counts with several batches in flight); same K order per output either way (bit-identical) */
    STEP_OPT_THROUGHPUT,       /*  0 (default): launch shapes tuned for the LATENCY of one batch | 1: for several independent batches in flight on separate streams (a serving loop) --
                                    a block's pointwise conv is launched on its own instead of riding in the 3x3x3 members' grid (the other batch fills the idle CUs; measured
                                    C2 +1.3 % at two in flight, -2.2 % one batch at a time).  Same bits either way */
    STEP_OPT_CONV_PERSIST,     /*  1 (default): one-channel-group conv_tap launches of more than one round of the chip run as a PERSISTENT tile loop, one workgroup per CU
                                    (the weight ring never drains, the next tile's halo is requested inside the current tile's epilogue) where the library has that
                                    form (the fused conv3d_2b -> conv3d_2c -> maxPool3d_3a call) | 0: one workgroup per tile (bit-identical) */
    STEP_OPT_CONV_PWS_WAVES,   /*  0 (default): the weight-stationary pointwise stream runs sixteen waves per workgroup where that gives every wave at most ONE 32-pixel
                                    group and eight waves do not (the 28x28 maps of 8 clips: 3136 groups), outside the throughput profile; eight otherwise | 8 | 16
                                    (bit-identical) */
    STEP_OPT_COUNT_
};
STEP_API int step_set_option(int option, int value);
STEP_API int step_get_option(int option, int* value);
STEP_API void step_reset_options(void);
STEP_API const char* step_option_name(int option);   /* "conv_impl", ... ; NULL for an unknown id */

/* ------------------------------------------------------------------------------------------
 * ROIAlign forward.     replaces _C.roi_align_forward   (csrc/ROIAlign.h:35-48,
 *                       cuda/ROIAlign_cuda.cu:88-146,281-323; cpu/ROIAlign_cpu.cpp:137-281)
 * feat  [B,C,H,W] (STEP_NCHW) or [B,H,W,C] (STEP_NHWC), dtype `dtype`
 * rois  [K,5] fp32 rows (batch_index_as_float, x1, y1, x2, y2) in input-image pixels
 * out   [K,C,ph,pw] (NCHW) or [K,ph,pw,C] (NHWC), same dtype as feat
 * "3-D ROIAlign over tubes" = this call on feat.view(B*T,...) with rois = tubes.view(-1,5)
 * (models/networks.py:42-45).
 */
STEP_API int step_roi_align_forward(const void* feat, int dtype, int layout, const float* rois, int K, int B,
                                    int C, int H, int W, int pooled_h, int pooled_w, float spatial_scale,
                                    int sampling_ratio, void* out, step_stream_t stream);

/* "3-D ROIAlign over tubes" without the caller's copy (utils/utils.py:41-48: `conv_feat[:, T_start:T_start+T_length].contiguous()`):
 * feat points at frame T_start of clip 0 inside a channels-last buffer [B, T_all, H, W, C]; a roi's first column is the frame index
 * b * T + t of the SLICE (flatten_tubes, utils/tube_utils.py:237-241) and is read as frame b * T_all + t of the buffer.  Same
 * arithmetic as step_roi_align_forward (bit-identical to it on the copied slice); out [K, ph, pw, C]. */
STEP_API int step_roi_align_tubes_forward(const void* feat, int dtype, const float* rois, int K, int B, int T_all, int T, int C,
                                          int H, int W, int pooled_h, int pooled_w, float spatial_scale, int sampling_ratio,
                                          void* out, step_stream_t stream);

/* ROIAlign backward.    replaces _C.roi_align_backward  (csrc/ROIAlign.h:51-69,
 *                       cuda/ROIAlign_cuda.cu:201-278,326-370)
 * grad [K,C,ph,pw] / [K,ph,pw,C]  ->  grad_feat [B,C,H,W] / [B,H,W,C] (fp32 only).
 * `mode` is a PER-CALL argument (it changes the fp32 summation order, i.e. results in the last bits, so it is not a process-wide
 * planner option: two threads of one process -- nn.DataParallel replicas -- cannot flip each other's determinism):
 *   STEP_ROI_BWD_GATHER  every cell of grad_feat gathers its samples in a fixed order, rois ascending (the same products
 *                        gtop * w / count as the reference's scatter): bit-reproducible, no clear, no atomics.  Every cell walks the
 *                        K roi headers (batch index first), so its cost grows with K x cells; the channels-last form evaluates a
 *                        roi's weights once per cell for all channels, the NCHW form once per (cell, channel).
 *   STEP_ROI_BWD_ATOMIC  the reference's algorithm: grad_feat is zeroed (ROIAlign_cuda.cu:340) and every sample adds its four
 *                        products with fp32 atomics (ROIAlign_cuda.cu:201-278); summation order varies from run to run.
 */
#define STEP_ROI_BWD_GATHER 0
#define STEP_ROI_BWD_ATOMIC 1
STEP_API int step_roi_align_backward(const float* grad, int layout, const float* rois, int K, int B, int C, int H,
                                     int W, int pooled_h, int pooled_w, float spatial_scale, int sampling_ratio,
                                     int mode, float* grad_feat, step_stream_t stream);

/* ROIPool forward.      replaces _C.roi_pool_forward    (csrc/ROIPool.h:35-48,
 *                       cuda/ROIPool_cuda.cu:40-101,134-180)
 * argmax: int32, same shape/layout as out; value h*W+w of the winning cell, -1 for empty bins.
 */
STEP_API int step_roi_pool_forward(const void* feat, int dtype, int layout, const float* rois, int K, int B, int C,
                                   int H, int W, int pooled_h, int pooled_w, float spatial_scale, void* out,
                                   int32_t* argmax, step_stream_t stream);

/* ROIPool backward.     replaces _C.roi_pool_backward   (csrc/ROIPool.h:50-69,
 *                       cuda/ROIPool_cuda.cu:103-132,183-226).  fp32, atomics. */
STEP_API int step_roi_pool_backward(const float* grad, const int32_t* argmax, int layout, const float* rois, int K,
                                    int B, int C, int H, int W, int pooled_h, int pooled_w, float* grad_feat,
                                    step_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * NMS, batched over independent groups.   replaces _C.nms   (csrc/nms.h:34-52)
 * Semantics are those of the reference CPU operator every reference script actually reaches
 * (cpu/nms_cpu.cpp:29-89; test.py:158-161 moves boxes to the CPU first): greedy, "+1" areas,
 * suppress when IoU >= threshold, score ties broken by lower index, bit-exact fp32 arithmetic.
 *   boxes  [G,kmax,4] fp32 (x1,y1,x2,y2)    scores [G,kmax] fp32    counts [G] int32 (<= kmax)
 *   keep   [G,kmax] uint8, 1 where the box survives (indices ascending == nonzero(keep))
 * A single nms(dets, scores, thr) call is G = 1.  "nms_3d" over tubes = one group per
 * (clip, class) holding the middle-frame boxes (test.py:174-195).
 *   scratch: step_nms_scratch_bytes(G,kmax) bytes (may be NULL when that returns 0).
 */
STEP_API size_t step_nms_scratch_bytes(int G, int kmax);
STEP_API int step_nms_batched(const float* boxes, const float* scores, const int32_t* counts, int G, int kmax,
                              float threshold, uint8_t* keep, void* scratch, step_stream_t stream);
/* The same for double-precision boxes / scores: the reference operator dispatches on the dtype (AT_DISPATCH_FLOATING_TYPES,
 * cpu/nms_cpu.cpp:95), so fp64 inputs are compared in fp64 -- down-casting them would move borderline IoU >= threshold decisions. */
STEP_API int step_nms_batched_f64(const double* boxes, const double* scores, const int32_t* counts, int G, int kmax,
                                  float threshold, uint8_t* keep, void* scratch, step_stream_t stream);

/* The evaluation loop of one refinement iteration (test.py:157-198; the same code in train.py:512-573, demo.py:123-174) in ONE
 * launch: for every clip b and class c, the clip's tubes whose middle-frame score prob[tube][c] > conf_thresh (test.py:180), their
 * middle-frame boxes through valid_tubes (utils/tube_utils.py:59-92: clamp to [0,width] x [0,height]; boxes under 3 px in either
 * direction become the whole frame), greedy NMS among them with the semantics of step_nms_batched (the reference compacts the
 * masked boxes first, order kept, so ties go to the lower tube).
 *   prob  [N, >= NC] fp32, row stride prob_stride elements     loc [N, >= 4] fp32, row stride loc_stride (x1,y1,x2,y2 in pixels)
 *   tube_start / tube_count [B] int32: clip b owns tubes tube_start[b] .. + tube_count[b] (<= kmax <= 64; more: STEP_E_UNSUPPORTED)
 *   keep [B, NC, kmax] uint8: 1 at the ORIGINAL slot of every surviving tube      boxes_out [N,4] (optional): the clamped boxes
 */
STEP_API int step_detect_nms(const float* prob, long long prob_stride, int NC, const float* loc, long long loc_stride,
                             const int32_t* tube_start, const int32_t* tube_count, int B, int kmax, float conf_thresh,
                             float nms_thresh, float width, float height, uint8_t* keep, float* boxes_out, step_stream_t stream);

/* The rows the evaluation loop appends behind the NMS (test.py:196-204), for up to STEP_DETECT_ITERS_MAX refinement iterations and all clips in
 * ONE launch: keep [I, B, NC, kmax] (step_detect_nms' masks), boxes[i] [N,4] (its clamped boxes) and scores[i] [N, >= NC] (row stride
 * score_strides[i]) of iteration i -- `boxes`, `scores`, `score_strides` are HOST arrays of I entries.  Group g = i * B + b owns the rows
 * [g * cap, g * cap + counts[g]) of the outputs, cap = NC * kmax, in the reference's order (classes ascending, kept tubes in ascending
 * original order): out_boxes [I*B*cap, 4] = box / [W,H,W,H] (test.py:197-198), out_scores, out_cls (class index), out_tube (index of the
 * tube inside its clip), counts [I*B] int32. */
#define STEP_DETECT_ITERS_MAX 8
STEP_API int step_detect_compact(const uint8_t* keep, const float* const* boxes, const float* const* scores, const long long* score_strides,
                                 const int32_t* tube_start, int I, int B, int NC, int kmax, float width, float height, float* out_boxes,
                                 float* out_scores, long long* out_cls, long long* out_tube, int32_t* counts, step_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Fused convolution unit on channels-last activations:
 *     y = act( conv(x, w) * scale[c] + shift[c] (+ residual) )
 * replaces Unit3Dpy = ConstantPad3d + Conv3d + BatchNorm3d(eval) + ReLU (models/i3dpt.py:43-111),
 * the biased 1x1x1 convs and Linear layers of TwoBranchNet (models/two_branch.py:182-200) and the
 * 2-D Bottleneck convs (two_branch.py:60-111; a 2-D conv is the kd = 1 case with D = frames).
 *
 *  x  [N, D, H, W, x_cstride] channels-last; the conv reads channels [x_coff, x_coff+Cin)
 *  y  [N, Do, Ho, Wo, y_cstride]; the conv writes channels [y_coff, y_coff+Cout)  (this is how an
 *     Inception block's torch.cat (i3dpt.py:162) disappears: branches write channel slices)
 *  res (optional) same geometry as y with its own cstride/coff, added before the activation.
 *  Stride is 1 in every dimension and padding is TF-"SAME" = k/2 per side (i3dpt.py:14-40) for
 *  this entry point; the strided 7x7x7 stem has its own entry point below.
 *  Weights are pre-packed by step_conv_pack_weight into MFMA fragment order.
 *  scale/shift are fp32 [Cout]: folded eval-mode BN (scale = gamma/sqrt(var+eps),
 *  shift = beta - mean*scale) or (1, bias) or NULL (=> 1, 0).
 */
typedef struct step_conv_desc {
    int dtype;                 /* STEP_F32 | STEP_BF16 | STEP_F16 (activations and packed weights) */
    int N, D, H, W;            /* input (= output) spatial extent */
    int Cin, Cout;
    int kd, kh, kw;            /* 1x1x1, 3x3x3 or 1x3x3 */
    int x_cstride, x_coff;
    int y_cstride, y_coff;
    int res_cstride, res_coff; /* used when res != NULL */
    int relu;
    /* optional second destination (1x1x1 convs only): output channels [split, Cout) are written to y2 at
     * channel (co - split) + y2_coff of a [N,D,H,W,y2_cstride] buffer instead of y.  split = 0 => unused.
     * This is how the three 1x1x1 convs of an Inception block that read the same input (branch_0 and the
     * two bottlenecks, i3dpt.py:133-148) run as ONE GEMM: branch_0 lands in the block's output slice, the
     * bottlenecks in the scratch buffer the 3x3x3 convs read. */
    int split;
    int y2_cstride, y2_coff;
} step_conv_desc;

/* number of ELEMENTS (of dtype) of the packed weight for a conv [Cout,Cin,kd,kh,kw] */
STEP_API size_t step_conv_packed_elems(int Cout, int Cin, int kd, int kh, int kw);
/* pack torch-layout weights  w[Cout][Cin][kd][kh][kw] (fp32, DEVICE) into fragment order, casting
 * to `dtype`.  `perm_c` (optional, device int32 [Cin]) remaps input channels: packed channel c reads
 * w[:, perm_c[c]] -- used to fold the NCHW->NHWC flatten order of Linear / global_cls weights
 * (two_branch.py:239-240,262) into the weights once. */
STEP_API int step_conv_pack_weight(const float* w, int Cout, int Cin, int kd, int kh, int kw, int dtype,
                                   const int32_t* perm_c, void* packed, step_stream_t stream);
/* The packed weight of the DATA-GRADIENT conv, straight from the forward weight w[Cout][Cin][kd][kh][kw] (fp32, DEVICE; odd
 * kernels): the conv gy [.., cin_pad] -> gx [.., Cin] of Conv3d/Conv2d/Linear.backward uses w with the channel roles swapped
 * and every tap axis flipped; cin_pad >= Cout is the gradient's channel count after the caller's padding to 16-byte vectors
 * (the extra input channels get zero weights).  Size: step_conv_packed_elems(Cin, cin_pad, kd, kh, kw).  Same image as
 * step_conv_pack_weight on the flipped / transposed / padded tensor, without materialising it. */
STEP_API int step_conv_pack_weight_dgrad(const float* w, int Cout, int Cin, int kd, int kh, int kw, int dtype, int cin_pad,
                                         void* packed, step_stream_t stream);
/* Every conv weight of a net re-packed by ONE launch (a training step changes all of them at once: ~290 separate pack
 * launches per step otherwise).  `items` is a DEVICE array of n descriptors.  An item with dgrad = 0 is
 * step_conv_pack_weight of the effective weight  w[:, cin_lo + perm_c[j]]  (j < Cin; perm_c NULL = identity) of the
 * parameter w[Cout][w_cin][taps]; an item with dgrad = 1 is step_conv_pack_weight_dgrad of that effective weight
 * (Cout = the FORWARD conv's output channels, Cin = its effective input channels, cin_pad >= Cout).  Results are bit-identical
 * to the single-weight entries. */
typedef struct step_pack_item {
    const float* w;          /* the parameter, torch layout [Cout][w_cin][taps], fp32, device */
    const int32_t* perm_c;   /* optional device int32 [Cin] */
    void* packed;            /* destination (dtype of the call) */
    int Cout, Cin, w_cin, cin_lo;
    int kd, kh, kw;
    int dgrad, cin_pad;      /* cin_pad: dgrad items only */
    int reserved;
} step_pack_item;
STEP_API int step_conv_pack_weights(const step_pack_item* items, int n, int dtype, step_stream_t stream);
STEP_API int step_conv_forward(const step_conv_desc* d, const void* x, const void* w_packed, const float* scale,
                               const float* shift, const void* res, void* y, void* y2, step_stream_t stream);

/* Several INDEPENDENT convs as one launch where the planner can give them one kernel instantiation -- today: two 16-bit 3x3x3
 * layers on the two-phase conv_tap kernel with general tile boxes, i.e. an Inception block's branch_1 / branch_2 3x3x3 convs
 * (models/i3dpt.py:141-150), which read different slices of the bottleneck buffer and write different slices of the block output.
 * Alone neither fills the chip's last round and every launch boundary idles all CUs for a prologue + an epilogue; in one grid the
 * narrow conv's workgroups run on the CUs the wide one leaves idle.  Results are bit-identical to n step_conv_forward calls (same
 * tiles, same K order); groups the planner cannot merge are launched one after the other.  No member may use `split`; outputs must
 * not overlap any member's input.
 * A third item may be a pointwise (1x1x1) conv -- the block's branch_3 conv on the pooled tensor: its workgroups are appended to the
 * same grid and run on the CUs the 3x3x3 members leave idle or free first (the 14x14 maps: 168-256 one-per-CU workgroups of very
 * different lengths) instead of as a 9-20 us launch of their own (option conv_group_pw: the workgroup limit, 0 = always separate;
 * bit-identical). */
/* An Inception block's max pool (3x3x3, stride 1, TF SAME: models/i3dpt.py:151-155) and a pointwise conv (the block's fused 1x1x1
 * triple, `split` allowed) as ONE launch: both read the block input, neither fills the chip on small maps, and in one grid the
 * workgroups of the two run side by side instead of one launch waiting for the other.  pool: x [N,D,H,W,C] -> pool_y (same shape);
 * conv: step_conv_forward(d, cx, ...) semantics.  Results are bit-identical to step_maxpool3d_tf + step_conv_forward.  16-bit
 * storage and pointwise layers the planner streams only; otherwise STEP_E_UNSUPPORTED (the caller launches the two separately). */
STEP_API int step_pool_conv_forward(int dtype, const void* x, int N, int D, int H, int W, int C, int x_cstride, int x_coff, void* pool_y,
                                    int py_cstride, int py_coff, const step_conv_desc* d, const void* cx, const void* w_packed,
                                    const float* scale, const float* shift, void* y, void* y2, step_stream_t stream);

/* What step_pool_conv_forward would do with the conv d: 0 = no combined form (the caller launches pool and conv separately), 1 | 2 = the
 * accumulator depth of its pointwise workgroups (64 | 128 output channels each; the kernel's second template argument). */
STEP_API int step_pool_conv_plan_nb(const step_conv_desc* d);

/* step_conv_forward over the channel CONCAT of two tensors that is never materialised (round 6): the reference's resample Bottleneck
 * applies conv1 / conv2 to torch.cat((global_feat, downsampled), 1) (models/two_branch.py:86-111, 313-319); here d->Cin = cin_a + cin_b,
 * channels [0, cin_a) are read from x (d's x_cstride / x_coff) and channels [cin_a, d->Cin) from xb (xb_cstride / xb_coff), same pixels.
 * One fp32 accumulation over the whole K, as the reference's conv over the concat has (two accumulating launches round the partial
 * sum to the storage type in between).  The form that exists: 16-bit storage, pointwise layers the planner streams (conv_pw_kernel
 * shapes), cin_a % 32 == 0, cin_b / xb_cstride / xb_coff % 8 == 0, xb 16-byte aligned; otherwise STEP_E_UNSUPPORTED and the caller
 * launches the halves one after the other (step_conv_forward with res = the first half's output). */
STEP_API int step_conv_forward_cat(const step_conv_desc* d, const void* x, int cin_a, const void* xb, int xb_cstride, int xb_coff,
                                   const void* w_packed, const float* scale, const float* shift, const void* res, void* y, void* y2,
                                   step_stream_t stream);

/* step_conv_forward with its INPUT produced on the fly: y = conv3x3x3(relu(pre_scale * conv1x1x1(x, pre_w) + pre_shift)) -- the pair
 * conv3d_2b_1x1 -> conv3d_2c_3x3 of the backbone (models/i3dpt.py:207-209) without the tensor between them.  x [N,D,H,W,pre_cin]
 * (d->x_cstride / x_coff describe it), pre_w_packed = step_conv_pack_weight of the [d->Cin, pre_cin, 1,1,1] weight, d->Cin = its
 * output channels.  The pointwise layer is evaluated per halo pixel while the 3x3x3 kernel stages its input tile (same K order and
 * the same 16-bit rounding as the separate launch: results are bit-identical to step_conv_forward twice).  Today: 16-bit storage,
 * pre_cin = d->Cin = 64, layers the planner sends to the two-phase conv_tap form; anything else returns STEP_E_UNSUPPORTED and the
 * caller launches the two layers separately. */
STEP_API int step_conv_forward_pre(const step_conv_desc* d, const void* x, const void* w_packed, const float* scale, const float* shift,
                                   const void* pre_w_packed, const float* pre_scale, const float* pre_shift, int pre_cin, void* y,
                                   step_stream_t stream);

/* ... and with the (1,3,3) / (1,2,2) max pool BEHIND it taken on the conv's tiles while they are on the chip: the triple
 * conv3d_2b_1x1 -> conv3d_2c_3x3 -> maxPool3d_3a_3x3 (models/i3dpt.py:207-212: Unit3Dpy, Unit3Dpy, MaxPool3dTFPadding) as one call.
 * y_pooled [N, D, Hp, Wp, Cout] with Hp = step_pool_out_size(H, 3, 2) (d->y_cstride / y_coff describe IT); the un-pooled conv
 * output never exists.  Two launches of the conv kernel at most (full rounds + the partial last round) plus one that completes the
 * pooled pixels on tile seams from `ws` (caller-owned, step_conv_pre_pool_workspace_bytes(d) bytes, 16-byte aligned: the tiles'
 * first rows and columns).  Bit-identical to step_conv_forward_pre followed by step_maxpool3d_tf.  Supported where
 * step_conv_forward_pre is AND the planner tiles the map 4 planes x 8 x 8 (sides the 8-pixel tiles cover best: 56 x 56 at T=32,
 * 224^2), d->relu = 1 (the pooled epilogue orders 16-bit patterns as integers: values >= +0), channel strides / offsets on the
 * 16-byte grid; otherwise the workspace size is 0 and the call returns STEP_E_UNSUPPORTED (the caller keeps the two calls). */
STEP_API size_t step_conv_pre_pool_workspace_bytes(const step_conv_desc* d);
STEP_API int step_conv_forward_pre_pool(const step_conv_desc* d, const void* x, const void* w_packed, const float* scale, const float* shift,
                                        const void* pre_w_packed, const float* pre_scale, const float* pre_shift, int pre_cin, void* y_pooled,
                                        void* ws, size_t ws_bytes, step_stream_t stream);
/* The same call in its two parts, for callers that account the launches separately (bench.py's per-kernel roofline): _tiles = the conv
 * launches (pooled tiles + the tiles' first rows / columns into ws), _finish = the seam pass over y_pooled.  _tiles then _finish on one
 * stream == step_conv_forward_pre_pool. */
STEP_API int step_conv_forward_pre_pool_tiles(const step_conv_desc* d, const void* x, const void* w_packed, const float* scale, const float* shift,
                                              const void* pre_w_packed, const float* pre_scale, const float* pre_shift, int pre_cin, void* y_pooled,
                                              void* ws, size_t ws_bytes, step_stream_t stream);
STEP_API int step_conv_pre_pool_finish(const step_conv_desc* d, void* y_pooled, void* ws, size_t ws_bytes, step_stream_t stream);

typedef struct step_conv_item {
    const step_conv_desc* desc;
    const void* x; const void* w_packed; const float* scale; const float* shift; const void* res; void* y;
} step_conv_item;
STEP_API int step_conv_forward_group(const step_conv_item* items, int n, step_stream_t stream);
/* Diagnostic: the kernel name of the merged launch, or "" when step_conv_forward_group would launch the members separately. */
STEP_API int step_conv_group_kernel_name(const step_conv_item* items, int n, char* buf, int buflen);

/* Same, with a caller-owned scratch buffer.  Few-row / very-deep-K pointwise layers (the heads' Linear(12544 -> 60 / 12),
 * models/two_branch.py:196,209-211) are split along K over the whole chip when `ws` holds at least
 * step_conv_workspace_bytes(d) bytes (16-byte aligned; contents are scratch, no initialisation needed); with ws = NULL
 * or for every other layer this is step_conv_forward.  The library never allocates: the caller owns ws. */
STEP_API size_t step_conv_workspace_bytes(const step_conv_desc* d);
STEP_API int step_conv_forward_ws(const step_conv_desc* d, const void* x, const void* w_packed, const float* scale,
                                  const float* shift, const void* res, void* y, void* y2, void* ws, size_t ws_bytes,
                                  step_stream_t stream);

/* Weight gradient of the same conv (training, train.py:257-348; replaces the cuDNN wgrad behind Conv3d/Conv2d/Linear
 * .backward):  dw[co][ci][kd][kh][kw] (fp32, torch's weight layout) (+)= sum_p dy[p][co] * x[p + tap][ci].
 * x is the forward input described by d (x_cstride / x_coff, dtype), dy the fp32 gradient w.r.t. the conv output
 * BEFORE the affine epilogue, laid out [N,D,H,W,y_cstride] at y_coff.  accumulate = 0 zero-fills dw first.
 * Different wavefronts meet in fp32 atomics: results are deterministic up to fp32 summation order. */
STEP_API int step_conv_wgrad(const step_conv_desc* d, const void* x, const float* dy, float* dw, int accumulate,
                             step_stream_t stream);

/* The same with a caller-owned scratch buffer (step_conv_wgrad_workspace_bytes(d) bytes, 16-byte aligned, no initialisation needed):
 * every wavefront job writes its partial tile to `ws` and a second kernel sums a tile's jobs in a FIXED order -- no atomics (a job
 * used to end in 4096 of them), bit-reproducible run to run, and accumulate = 0 needs no clear.  ws = NULL or too small:
 * step_conv_wgrad.  The library never allocates: the caller owns ws. */
STEP_API size_t step_conv_wgrad_workspace_bytes(const step_conv_desc* d);
STEP_API int step_conv_wgrad_ws(const step_conv_desc* d, const void* x, const float* dy, float* dw, int accumulate, void* ws,
                                size_t ws_bytes, step_stream_t stream);

/* The same weight gradient on the 16-bit matrix instructions (mixed-precision training): dy arrives in the activation type
 * d->dtype (STEP_BF16 / STEP_F16; STEP_F32 -> STEP_E_UNSUPPORTED), laid out as above; products in 16 bits, fp32 accumulation,
 * fp32 dw.  16x the matrix rate of step_conv_wgrad (whose fp32 instruction runs at 1/16 of the 16-bit one). */
STEP_API int step_conv_wgrad16(const step_conv_desc* d, const void* x, const void* dy, float* dw, int accumulate,
                               step_stream_t stream);

/* Same, with a caller-owned scratch buffer: 3x3 windows (kh = kw = 3) run as an LDS-tiled GEMM whose workgroups write their
 * partial tiles to `ws` (step_conv_wgrad16_workspace_bytes(d) bytes, 16-byte aligned, no initialisation needed) and a second
 * kernel sums them in a fixed order -- no atomics, bit-reproducible; every other window / channel count runs the per-tap kernel with
 * the partial-tile scheme of step_conv_wgrad_ws (the workspace query covers both).  ws = NULL or a buffer that is too small:
 * step_conv_wgrad16.  The library never allocates: the caller owns ws. */
STEP_API size_t step_conv_wgrad16_workspace_bytes(const step_conv_desc* d);
STEP_API int step_conv_wgrad16_ws(const step_conv_desc* d, const void* x, const void* dy, float* dw, int accumulate, void* ws,
                                  size_t ws_bytes, step_stream_t stream);

/* Diagnostic: the name (as rocprofv3 prints it) of the kernel instantiation step_conv_forward launches
 * for this descriptor -- lets bench.py attribute time and algorithmic work to profiler rows. */
STEP_API int step_conv_kernel_name(const step_conv_desc* d, char* buf, int buflen);
/* Diagnostic: the launch plan of a descriptor as integers -- info[0..9] = implementation (0 tiled, 1 pipelined conv_tap, 2 streaming
 * pointwise, 3 split-K, 4 weight-stationary pointwise), log2 tile width (0 = general box), accumulator depth NB, waves per
 * workgroup, two-phase form, the general box (planes, rows, columns), its pixel assignment mode (1 = every 16-lane LDS read group is
 * one run of 16 columns of one box row), pixel tiles.  n >= 10. */
STEP_API int step_conv_plan_info(const step_conv_desc* d, int* info, int n);
/* ... and the plan step_conv_forward_pre_pool REALLY uses for d (it re-plans a general-box layer onto the 4 x 8 x 8 tiles when those are
 * at most 25 % more): info[0..9] as above, info[10..11] = tile rows / tile columns per plane (what the seam pass walks); with n >= 13
 * info[12] = workgroups of the persistent tile loop its NB = 3 launch runs as (conv_tap_pre_pool_persist_kernel; 0 = one workgroup
 * per tile).  n >= 12; STEP_E_UNSUPPORTED where the fused call has no form for d. */
STEP_API int step_conv_pre_pool_plan_info(const step_conv_desc* d, int* info, int n);

/* The I3D stem: 7x7x7 stride-2 conv, Cin = 3, TF-SAME padding (2 front, 3 back) + BN + ReLU
 * (models/i3dpt.py:186-191) reading the clip in the reference's own input layout
 *   x [N, T, 3, H, W] (what BaseNet.forward receives, models/networks.py:69-77)
 * and writing channels-last y [N, To, Ho, Wo, Cout] with To = ceil(T/2) etc.  relu = 0: the affine output without the ReLU (the
 * batch-statistics BatchNorm of --freeze_stats False follows as its own pass, step_bn_train_forward). */
STEP_API size_t step_stem_packed_elems(int Cout);
STEP_API int step_stem_pack_weight(const float* w /*[Cout,3,7,7,7]*/, int Cout, int dtype, void* packed,
                                   step_stream_t stream);
STEP_API int step_stem_forward(int dtype, const void* x, int N, int T, int H, int W, const void* w_packed,
                               const float* scale, const float* shift, int relu, int Cout, void* y, int y_cstride,
                               int y_coff, step_stream_t stream);
/* Weight gradients whose fixed-order sum is DEFERRED: step_conv_wgrad_partial launches only the kernel that writes the partial tiles
 * into `ws` (as step_conv_wgrad_ws / step_conv_wgrad16_ws with dy16 = 0 / 1 would) and fills *item -- a host-side descriptor of the sum
 * that is still to run; step_wgrad_reduce_group then runs up to STEP_WGRAD_REDUCE_MAX such sums as ONE launch (an Inception block's six
 * weight gradients: one reduce launch instead of six).  item->kind == 0: nothing is pending (the form accumulated into dw itself).
 * ws must stay alive and untouched until the group launch has run.  Same arithmetic and summation order as the _ws entry points
 * (bit-identical results). */
/* Diagnostic, as step_conv_kernel_name: the (main) kernel a weight-gradient call launches for this descriptor (dy16: the 16-bit entry). */
STEP_API int step_conv_wgrad_kernel_name(const step_conv_desc* d, int dy16, char* buf, int buflen);
#define STEP_WGRAD_REDUCE_MAX 8
typedef struct step_wgrad_reduce_item {     /* filled by step_conv_wgrad_partial; opaque to the caller apart from kind == 0 (nothing pending) */
    const float* ws; float* dw;
    long long jobs;                          /* partial results along the pixel axis */
    int kind;                                /* 0 none | 1 per-tap kernel's tiles | 2 LDS-tiled kernel's tiles | 3 dense [Cout, Cin] slice images (pointwise pixel stream) */
    int gy, nbw, cot, cit, Cout, Cin, taps, accumulate;
    int pw;                                  /* kind 2: 0 = 3x3 windows, six waves | 1 = pointwise | 2 = 3x3 windows, twelve waves (tile order inside a workgroup's block) */
} step_wgrad_reduce_item;
STEP_API int step_conv_wgrad_partial(const step_conv_desc* d, const void* x, const void* dy, int dy16, float* dw, int accumulate, void* ws,
                                     size_t ws_bytes, step_wgrad_reduce_item* item, step_stream_t stream);
STEP_API int step_wgrad_reduce_group(const step_wgrad_reduce_item* items, int n, step_stream_t stream);

/* The stem AND maxPool3d_2a_3x3 behind it (models/i3dpt.py:186-196: Unit3Dpy 7x7x7 / 2, then MaxPool3dTFPadding (1,3,3) / (1,2,2)) as one
 * call: every 16x16 stem tile is max-pooled while it is still on the chip, y is the POOLED tensor [N, To, Hp, Wp, Cout] (Hp / Wp =
 * step_pool_out_size(Ho / Wo, 3, 2)) and the un-pooled stem output -- the largest tensor of the backbone -- never reaches memory.
 * Pooled pixels on tile seams lack one row / column of the neighbouring tile: tiles leave their first row and column in `ws` and a
 * second, small launch completes the seams.  Bit-identical to step_stem_forward (relu = 1) + step_maxpool3d_tf.  16-bit dtypes,
 * W % 4 == 0, Cout == 64, y 16-byte aligned with y_cstride / y_coff multiples of 8; otherwise STEP_E_UNSUPPORTED (the caller runs the
 * two layers one after the other).  ws: step_stem_pool_workspace_bytes() bytes (0 = unsupported), 16-byte aligned. */
STEP_API size_t step_stem_pool_workspace_bytes(int dtype, int N, int T, int H, int W, int Cout);
STEP_API int step_stem_pool_forward(int dtype, const void* x, int N, int T, int H, int W, const void* w_packed, const float* scale,
                                    const float* shift, int Cout, void* y, int y_cstride, int y_coff, void* ws, size_t ws_bytes,
                                    step_stream_t stream);
/* The call in its two parts (what step_stem_pool_forward does in one): _tiles = the stem's launch (pooled tiles + their first rows / columns into
 * ws), _finish = the seam pass over y and ws.  Same arguments and checks (finish: x is only checked, not read); callers that time the two
 * launches separately (bench.py's per-kernel roofline) use these. */
STEP_API int step_stem_pool_forward_tiles(int dtype, const void* x, int N, int T, int H, int W, const void* w_packed, const float* scale,
                                          const float* shift, int Cout, void* y, int y_cstride, int y_coff, void* ws, size_t ws_bytes,
                                          step_stream_t stream);
STEP_API int step_stem_pool_finish(int dtype, const void* x, int N, int T, int H, int W, int Cout, void* y, int y_cstride, int y_coff, void* ws,
                                   size_t ws_bytes, step_stream_t stream);
/* The same call reading the decoder's uint8 frames [N,T,H,W,3] directly (device memory, 4-byte aligned): step_clip_from_u8's
 * normalisation -- scale 0 / 1 / 2, then (v - mean[c]) / std[c], data/augmentations.py:68-111 -- and the rounding to `dtype` happen
 * while the stem stages its frames (a 3 x 256-entry table built per workgroup with the same fp32 operations), so the normalised
 * clip never exists: half the input bytes and one pass less on a FED node.  Bit-identical to step_clip_from_u8 (into `dtype`) +
 * step_stem_pool_forward.  mean3 / std3: HOST pointers to 3 floats (NULL = 0 / 1).  Same support and workspace as
 * step_stem_pool_forward. */
STEP_API int step_stem_pool_forward_u8(int dtype, const unsigned char* frames, int N, int T, int H, int W, int scale_mode, const float* mean3,
                                       const float* std3, const void* w_packed, const float* scale, const float* shift, int Cout, void* y,
                                       int y_cstride, int y_coff, void* ws, size_t ws_bytes, step_stream_t stream);
/* Weight gradient of the stem: dw[Cout][3][7][7][7] (fp32, torch layout) (+)= sum over output pixels of
 * dy[n,to,ho,wo,co] * x_padded[...]; x as in step_stem_forward, dy fp32 contiguous [N,To,Ho,Wo,Cout] (gradient before
 * the affine epilogue).  The stem needs no data gradient (its input is the clip). */
STEP_API int step_stem_wgrad(int dtype, const void* x, int N, int T, int H, int W, const float* dy, int Cout, float* dw,
                             int accumulate, step_stream_t stream);
/* step_stem_wgrad with a caller-owned scratch (step_stem_wgrad_workspace_bytes; 16-byte aligned, no initialisation): every job
 * writes its partial tile and one fixed-order sum writes dw -- no fp32 atomics, bit-reproducible.  ws == NULL: the atomics form;
 * a scratch that is too short or misaligned is STEP_E_SHAPE. */
STEP_API size_t step_stem_wgrad_workspace_bytes(int N, int T, int H, int W, int Cout);
STEP_API int step_stem_wgrad_ws(int dtype, const void* x, int N, int T, int H, int W, const float* dy, int Cout, float* dw,
                                int accumulate, void* ws, size_t ws_bytes, step_stream_t stream);
/* The same gradient on the 16-bit matrix instructions (bf16 / fp16 clip; dy in the SAME 16-bit type, channels-last contiguous --
 * the activation gradient mixed-precision training back-propagates).  One workgroup per CU keeps all 49 x 21 filter columns of a
 * 32-channel block in registers and reads x and dy once; the partial tiles go through the caller-owned scratch `ws`
 * (step_stem_wgrad16_workspace_bytes; 16-byte aligned, no initialisation) and are summed in a fixed order: deterministic, no
 * atomics.  Needs W % 8 == 0 and Cout % 8 == 0 (16-byte rows); the workspace query returns 0 and the call STEP_E_UNSUPPORTED
 * for other shapes and for fp32, which stay with step_stem_wgrad. */
STEP_API size_t step_stem_wgrad16_workspace_bytes(int dtype, int N, int T, int H, int W, int Cout);
STEP_API int step_stem_wgrad16(int dtype, const void* x, int N, int T, int H, int W, const void* dy, int Cout, float* dw,
                               int accumulate, void* ws, size_t ws_bytes, step_stream_t stream);
/* Diagnostic, as step_conv_kernel_name: the kernel step_stem_forward launches for this dtype. */
STEP_API int step_stem_kernel_name(int dtype, char* buf, int buflen);

/* ------------------------------------------------------------------------------------------
 * TF-"SAME" max pool on channels-last activations.  replaces MaxPool3dTFPadding =
 * ConstantPad3d(0) + MaxPool3d(ceil_mode=True) (models/i3dpt.py:114-126): the explicit TF pad
 * (max(k-s,0), split floor/ceil, back-heavy) carries the VALUE 0; positions beyond it that a
 * ceil-mode window overhangs are ignored.
 *  x [N,D,H,W,x_cstride] channels [x_coff, x_coff+C)  ->  y [N,Do,Ho,Wo,y_cstride] at y_coff
 *  Do/Ho/Wo are given by step_pool_out_size(L, k, s).
 */
STEP_API int step_pool_out_size(int L, int k, int s);
STEP_API int step_maxpool3d_tf(int dtype, const void* x, int N, int D, int H, int W, int C, int x_cstride,
                               int x_coff, int kd, int kh, int kw, int sd, int sh, int sw, void* y, int y_cstride,
                               int y_coff, step_stream_t stream);

/* Backward of step_maxpool3d_tf (training): gx[N,D,H,W,C] (fp32, contiguous, zero-filled by the call) receives
 * gy[N,Do,Ho,Wo,C] (fp32, contiguous) at the first maximum of each window in (d,h,w) scan order -- torch's MaxPool3d
 * rule on the explicitly zero-padded tensor; a window won by a pad element drops its gradient. */
STEP_API int step_maxpool3d_tf_backward(int dtype, const void* x, int N, int D, int H, int W, int C, int x_cstride,
                                        int x_coff, int kd, int kh, int kw, int sd, int sh, int sw, const float* gy,
                                        float* gx, step_stream_t stream);
/* The same gradient as two gathers instead of fp32 atomics: pass A writes one byte per output element (the window tap of the
 * first maximum) into `arg_scratch` (N*Do*Ho*Wo*C bytes, device), pass B lets every input element collect the gradients of the
 * windows it won, in a fixed order (bit-reproducible), and writes gx once -- in fp32 or in the activation type, from gy in fp32 or
 * in the activation type (gy_dtype / gx_dtype: STEP_F32 or `dtype`), so a 16-bit net needs no fp32 staging, no clear and no
 * conversion pass.  gy dense [N,Do,Ho,Wo,C], gx dense [N,D,H,W,C].  C (and the slice of x) must be whole 16-byte channel vectors of
 * `dtype`, else STEP_E_UNSUPPORTED (the atomic entry above has no such limit).
 * Round 4: for 16-bit activations the (3,3,3) / (1,1,1) window (the Inception blocks' pool) runs as ONE launch that finds the first
 * maximum separably on LDS tiles and routes the gradient back the same way; it does not touch arg_scratch (still required non-NULL)
 * and adds the same terms in a different fixed order (planes, rows, columns) -- equal to the two-gather form up to fp32 rounding,
 * bit-reproducible run to run; STEP_OPT_POOL_DIRECT = 1 keeps the two gathers. */
STEP_API int step_maxpool3d_tf_backward_gather(int dtype, const void* x, int N, int D, int H, int W, int C, int x_cstride, int x_coff,
                                               int kd, int kh, int kw, int sd, int sh, int sw, int gy_dtype, const void* gy, int gx_dtype,
                                               void* gx, unsigned char* arg_scratch, step_stream_t stream);

/* Average pool over a full (kh x kw) window, stride 1, no padding ("VALID"), kd = 1.
 * replaces nn.AvgPool3d((1,13,13),(1,1,1)) of ContextNet (models/two_branch.py:127,136).
 * x [N,D,H,W,C] -> y [N,D,H-kh+1,W-kw+1,C] */
STEP_API int step_avgpool_hw(int dtype, const void* x, int N, int D, int H, int W, int C, int kh, int kw, void* y,
                             step_stream_t stream);

/* Layout / dtype conversion between the reference's NCDHW-style logical tensors and channels-last.
 *  src [N, C, S] (S = D*H*W flattened, i.e. torch-contiguous NCDHW)  <->  dst [N, S, C]
 *  to_channels_last = 1: NCS -> NSC ; 0: NSC -> NCS.  src/dst dtypes may differ (cast). */
STEP_API int step_transpose_cs(const void* src, int src_dtype, void* dst, int dst_dtype, int N, int C, long long S,
                               int to_channels_last, step_stream_t stream);

/* Clip ingest: uint8 frames [N,T,H,W,3] (device memory; the decoder's layout, data/ava.py:298-338) -> the normalised
 * clip [N,T,3,H,W] in `dtype` that step_stem_forward / BaseNet.forward take.  Same fp32 arithmetic, in the same order, as
 * the reference's host-side ConvertFromInts(scale) + SubtractMeans + DivideStds (data/augmentations.py:68-111,600-612):
 * scale 0: x, 1: x/255, 2: x*2/255 - 1; then (v - mean[c]) / std[c].  mean3 / std3 are HOST pointers to 3 floats (NULL = 0 / 1).
 * Frames must already have the network's resolution (the reference resizes on the host between the two steps). */
STEP_API int step_clip_from_u8(const unsigned char* frames, int N, int T, int H, int W, int scale, const float* mean3,
                               const float* std3, int dtype, void* clip, step_stream_t stream);

/* Fused multi-tensor Adam over flat fp32 arenas: replaces optimizer.step() of torch.optim.Adam(params, lr=args.det_lr)
 * (train.py:126,348) over the single-tensor parameter groups of utils/solver.py:12-93 (per-group lr / weight_decay; the
 * schedulers of solver.py:96-180 rewrite group['lr'] between steps).  param / grad / exp_avg / exp_avg_sq: n fp32 elements
 * each (n % 4 == 0, 16-byte aligned).  Segment s covers elements [seg_end[s-1], seg_end[s]) (seg_end ascending, every entry
 * a multiple of 4, seg_end[n_seg-1] == n) and uses seg_lr[s], seg_wd[s]; the three tables are DEVICE arrays, n_seg <= 4096.
 * Arithmetic = torch/optim/adam.py::_single_tensor_adam (amsgrad off): g = grad*grad_scale + wd*p; m += (g-m)(1-beta1);
 * v = v*beta2 + (1-beta2) g*g; p -= lr/(1-beta1^step) * m / (sqrt(v)/sqrt(1-beta2^step) + eps).  step counts from 1.
 * beta1 / beta2 / eps are doubles as in torch (1 - beta is taken in double).  grad_scale folds the 1/world_size of the gradient average (or 1/loss_scale) into the same pass; zero_grad != 0 clears the
 * gradient arena on the way out (optimizer.zero_grad(), train.py:287). */
STEP_API int step_adam_flat(float* param, float* grad, float* exp_avg, float* exp_avg_sq, long long n,
                            const long long* seg_end, const float* seg_lr, const float* seg_wd, int n_seg, double beta1,
                            double beta2, double eps, int step, float grad_scale, int zero_grad, step_stream_t stream);
/* The same update with the step counter ON THE DEVICE (torch.optim.Adam(capturable=True)): *step_dev (int64, device) is
 * incremented by the call and the bias corrections are derived from it on the device (bias_corr: 2 floats of device scratch), so
 * the launch can sit in a captured HIP graph and advance on every replay. */
STEP_API int step_adam_flat_dev(float* param, float* grad, float* exp_avg, float* exp_avg_sq, long long n,
                                const long long* seg_end, const float* seg_lr, const float* seg_wd, int n_seg, double beta1,
                                double beta2, double eps, long long* step_dev, float* bias_corr, float grad_scale, int zero_grad,
                                step_stream_t stream);
/* Mixed-precision training with dynamic loss scaling (train.py:136-139 apex amp O1, :342-345 `with amp.scale_loss(loss, optimizer)`),
 * as one call with every decision on the device (capturable): amp_state = 4 device floats {scale, growth_tracker, found_inf, unused}.
 * The caller multiplied the loss by amp_state[0] before backward, so `grad` holds scale x the gradients.  The call
 *   1. scans the gradient arena for inf / nan (found_inf = 1),
 *   2. runs step_adam_flat_dev with grad_scale / scale -- or, when found_inf is set, SKIPS the step as apex's patched
 *      optimizer.step() / torch.amp.GradScaler.step() do: parameters, moments and the step count stay as they are (zero_grad
 *      still clears the gradients),
 *   3. updates the scale like DynamicLossScaler / GradScaler.update(): overflow -> scale *= backoff_factor, tracker = 0; else
 *      tracker += 1 and after growth_interval clean steps scale *= growth_factor; found_inf = 0.
 * apex's defaults: initial scale 2^16, growth 2, backoff 0.5, interval 2000. */
STEP_API int step_adam_flat_amp(float* param, float* grad, float* exp_avg, float* exp_avg_sq, long long n,
                                const long long* seg_end, const float* seg_lr, const float* seg_wd, int n_seg, double beta1,
                                double beta2, double eps, long long* step_dev, float* bias_corr, float grad_scale, int zero_grad,
                                float* amp_state, float growth_factor, float backoff_factor, int growth_interval,
                                step_stream_t stream);

/* The tail of TwoBranchNet.forward (models/two_branch.py:246-342) behind its last two GEMMs, as ONE launch (and one for its backward):
 * the class logits averaged over a tube's frames and their sigmoid, the box regressions (local_loc = columns 0..3 of the fused
 * 12-column regressor, first_loc / last_loc = local_loc + columns 4..7 / 8..11 on the first / last chunk), and -- with targets -- the
 * three losses: BCE-with-logits against the centre frame's labels x mask (:281-299) and the masked-mean smooth-L1 of the centre /
 * first / last predictions against their encode_coef targets (:301-333, utils/tube_utils.py:127-157).  ~60 element-wise kernels per head
 * and step in the reference's formulation (and as many in backward).
 *   logits [N * Tl rows, >= NC], reg [N * Tl rows, >= 12] (NULL: cls_only) in `dtype`, row strides in elements
 *   tubes [N, Tl, 5] fp32, targets [N, 3, 6 + NC] fp32 (rows: first, centre, last frame; [4] class mask, [5] box mask, [6:] labels) --
 *   both NULL for inference (the three losses are then written as zeros)
 *   prob [N, NC], local_loc [N, Tl, 4], first_loc / last_loc [N, T, 4], loss_cls [N, NC] ([1] without targets), loss_loc [1], loss_nbr [1]: fp32
 * An all-zero mask yields exactly zero losses and gradients (the reference's `if mask.sum():` branches, taken on the device).
 * backward: g_loss_* = the gradients of the three loss outputs (NULL = zero) -> g_logits [N * Tl, NC], g_reg [N * Tl, 12] dense, `dtype`.
 * One 256-thread workgroup, fixed-order sums (bit-reproducible). */
STEP_API int step_head_outputs(int dtype, const void* logits, int logits_stride, const void* reg, int reg_stride, int N, int Tl, int T,
                               int NC, const float* tubes, const float* targets, float* prob, float* local_loc, float* first_loc,
                               float* last_loc, float* loss_cls, float* loss_loc, float* loss_nbr, step_stream_t stream);
STEP_API int step_head_outputs_backward(int dtype, const void* logits, int logits_stride, const void* reg, int reg_stride, int N, int Tl,
                                        int T, int NC, const float* tubes, const float* targets, const float* g_loss_cls,
                                        const float* g_loss_loc, const float* g_loss_nbr, void* g_logits, void* g_reg,
                                        step_stream_t stream);

/* Batch-statistics BatchNorm3d (+ ReLU) of a conv unit: the TRAINING-mode BatchNorm of --freeze_stats False (models/networks.py:85-99
 * leaves the layers in train mode; models/i3dpt.py:95-110, models/two_branch.py:160,372).  Eval-mode BN is folded into the conv
 * epilogues and never comes here.
 * forward:  z [M, C] channels-last (the raw conv output; z_cstride elements between pixels, 0 = C) ->
 *             y = relu?(gamma * (z - mean_B) / sqrt(var_B + eps) + beta)      mean_B / var_B (biased) over the M pixels of THIS batch
 *           save_mean / save_invstd [C] fp32 (for backward); running_mean / running_var (may be NULL) move by `momentum` towards the
 *           batch mean / UNBIASED variance, as torch.nn.BatchNorm3d(momentum=0.1) does.  gamma / beta NULL = 1 / 0.
 * backward: gy [M, C] (fp32 or `dtype`), y the forward's output (read for the ReLU mask only) ->
 *             gz [M, C] dense `dtype` (gradient w.r.t. the conv output), ggamma / gbeta [C] fp32 (may be NULL)
 * Fixed-order reductions (chunk partials in a caller-owned workspace, merged in double precision): bit-reproducible, no atomics.
 * ws: step_bn_train_workspace_bytes(M, C) bytes, 16-byte aligned.  C and the strides must be multiples of 4. */
STEP_API size_t step_bn_train_workspace_bytes(long long M, int C);
STEP_API int step_bn_train_forward(int dtype, const void* z, int z_cstride, long long M, int C, const float* gamma, const float* beta,
                                   float eps, float momentum, float* running_mean, float* running_var, float* save_mean,
                                   float* save_invstd, int relu, void* y, int y_cstride, void* ws, size_t ws_bytes, step_stream_t stream);
STEP_API int step_bn_train_backward(int dtype, const void* z, int z_cstride, const void* y, int y_cstride, int gy_dtype, const void* gy,
                                    int gy_cstride, long long M, int C, int relu, const float* gamma, const float* save_mean,
                                    const float* save_invstd, void* gz, float* ggamma, float* gbeta, void* ws, size_t ws_bytes,
                                    step_stream_t stream);

/* Activation gradient of the fused conv unit (the backward of Unit3Dpy's BatchNorm3d(eval) + ReLU, models/i3dpt.py:100-111,
 * and of the Bottleneck ReLUs, two_branch.py:60-111, which autograd runs as separate element-wise kernels in the reference):
 *   g[m][c] = gy[m][c] * (relu ? y[m][c] > 0 : 1) * (scale ? scale[c] : 1)      m < M pixels, c < C channels (channels-last)
 * written as fp32 into g32 (operand of step_conv_wgrad; may be NULL) and / or in `dtype` into g_act (operand of the
 * data-gradient step_conv_forward; may be NULL), both dense [M, C].  y / g_act have `dtype`; gy has gy_dtype = STEP_F32 or
 * `dtype`; y and gy may be channel slices of wider channels-last buffers (y_cstride / gy_cstride = elements between
 * consecutive pixels, 0 = C): the Inception branches write into and read their gradient from the concat buffer.
 * C or a stride not a multiple of 4 -> STEP_E_UNSUPPORTED (the caller keeps torch's element-wise path for those few layers). */
STEP_API int step_act_grad(int dtype, const void* y, int y_cstride, int gy_dtype, const void* gy, int gy_cstride, const float* scale,
                           long long M, int C, int relu, float* g32, void* g_act, step_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * One refinement step's tube bookkeeping (utils/utils.py:68-129 in temporal_mode "predict"), one launch:
 *   pred_loc   [N,T,4]  = decode_coef(tubes[:, :, 1:5],                         local_loc)      (tube_utils.py:178-189)
 *   pred_first [N,Tw,4] = decode_coef(tubes[:, first_off : first_off+Tw, 1:5],  first_loc)      (utils.py:75-79)
 *   pred_last  [N,Tw,4] = decode_coef(tubes[:, last_off  : last_off +Tw, 1:5],  last_loc)
 *   next_tubes [N,Tn,5] = the next step's proposals: extend ? cat(pred_first, pred_loc, pred_last) : pred_loc   (Tn = T + 2 Tw | T),
 *                         through valid_tubes(width, height) (clamp; boxes under 3 px become the whole frame, tube_utils.py:59-92),
 *                         with column 0 = clip_of[n] * Tn + frame (flatten_tubes(batch_idx=True), tube_utils.py:214-246).
 * tubes [N,T,5] fp32 (column 0 ignored), local_loc [N,T,4], first_loc / last_loc [N,Tw,4] fp32, clip_of [N] int32.
 * fp32 arithmetic in the reference's operation order. */
STEP_API int step_tube_update(const float* tubes, int N, int T, const float* local_loc, const float* first_loc,
                              const float* last_loc, int Tw, int first_off, int last_off, const int32_t* clip_of, int extend,
                              float width, float height, float* pred_loc, float* pred_first, float* pred_last,
                              float* next_tubes, step_stream_t stream);

/* Training sample selection, device front end (utils/utils.py:179-214 train_select, utils/tube_utils.py:59-92,269-351): from a
 * previous step's predictions -- prob [N,T,NC], loc [N,T,4], first / last [N,Tw,4] (NULL outside temporal_mode "predict"), all fp32
 * dense -- in one launch: mean_prob [N,NC] = the class scores averaged over the tube's frames (sequential fp32 sum / T, as numpy's
 * mean does), vloc / vfirst / vlast = the tubes through valid_tubes(width, height), and iou [N,Gmax] = box IoU (no +1 convention; 0
 * unless both overlap extents are positive; 0 for an all-zero padding box) of the tube's clamped middle-frame box with the
 * ground-truth boxes gt_mid [B,Gmax,4] of its clip (clip_of [N] int32, gt_count [B] int32).  The draws from the random streams and
 * the stable sorts that pick the training tubes stay on the host (step_amd/selection.py), fed by ONE small device-to-host copy. */
STEP_API int step_select_prepare(const float* prob, const float* loc, const float* first, const float* last, int N, int T, int Tw,
                                 int NC, const int32_t* clip_of, const float* gt_mid, const int32_t* gt_count, int Gmax,
                                 float width, float height, float* mean_prob, float* vloc, float* vfirst, float* vlast,
                                 float* iou, step_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* STEP_AMD_H */
