// step_amd/csrc/head.hip -- everything TwoBranchNet.forward does AFTER its last two GEMMs (models/two_branch.py:246-333), forward and
// backward, as one launch each:
//   global_class = mean over the tube's frames of the class logits (:247-249), global_prob = sigmoid (:341)
//   local_loc = the regressor's first 4 columns, first_loc / last_loc = local_loc + the neighbour regressors on the first / last
//   chunk (:262-270), center / first / last predictions (:271-273)
//   loss_global_cls = BCE-with-logits against the centre frame's labels, masked (:281-299)
//   loss_local_loc / loss_neighbor_loc = masked-mean smooth-L1 against encode_coef targets (:301-333; utils/tube_utils.py:127-157)
// In the reference (and in this package until round 4) that is ~60 element-wise torch kernels per head and step and as many again
// in backward -- ~350 launches of a 1300-launch training step, every one a few microseconds of latency on a few hundred values.
// The tensors are tiny (N <= a few hundred tubes, 60 classes, 12 regressor columns): ONE 256-thread workgroup does it all, with
// fixed-order reductions (bit-reproducible).  The reference's `if mask.sum():` host branches are taken on the device: an all-zero
// mask gives exactly zero losses and zero gradients (heads.py SYNC_FREE_LOSSES documents the one visible difference in shape).
#include "common.h"

namespace step {

struct HeadParams {
    const void* logits; int lcs;          // [N*Tl rows, >= NC] activation dtype, row stride lcs elements
    const void* reg; int rcs;             // [N*Tl rows, >= 12]
    int N, Tl, T, NC;
    int c_first, c_mid, c_last;           // frame of the first / middle / last chunk's centre (two_branch.py:226-228)
    int lo, lo2;                          // first frame of the first / last chunk
    const float* tubes;                   // [N, Tl, 5] or NULL (inference)
    const float* targets; int tstride;    // [N, 3, 6 + NC] fp32, tstride = 3 * (6 + NC); NULL: no losses
    float* prob; float* local_loc; float* first_loc; float* last_loc;      // [N,NC], [N,Tl,4], [N,T,4], [N,T,4]
    float* loss_cls; float* loss_loc; float* loss_nbr;                     // [N,NC], [1], [1]
    // backward
    const float* g_cls; const float* g_loc; const float* g_nbr;           // upstream gradients ([N,NC] | NULL, [1] | NULL, [1] | NULL)
    void* g_logits; void* g_reg;          // [N*Tl, NC], [N*Tl, 12] dense, activation dtype
};

template <typename T> __device__ __forceinline__ float ld(const void* p, size_t i) { return elem<T>::to_f32(((const T*)p)[i]); }
template <typename T> __device__ __forceinline__ void st(void* p, size_t i, float v) { ((T*)p)[i] = elem<T>::from_f32(v); }

// fixed-order block sum of one float per thread (256 threads): strided partials, then a tree in LDS
__device__ __forceinline__ float block_sum(float v, float* red) {
    const int tid = threadIdx.x;
    __syncthreads();
    red[tid] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    return red[0];
}

// encode_coef of one coordinate k (utils/tube_utils.py:139-157): gt box g[4], proposal a[4] (x1,y1,x2,y2), "+1" sizes
__device__ __forceinline__ float encode_k(const float* g, const float* a, int k) {
    const float gw = g[2] - g[0] + 1.0f, gh = g[3] - g[1] + 1.0f, w = a[2] - a[0] + 1.0f, h = a[3] - a[1] + 1.0f;
    switch (k) {
        case 0: return ((g[0] + 0.5f * gw) - (a[0] + 0.5f * w)) / w;
        case 1: return ((g[1] + 0.5f * gh) - (a[1] + 0.5f * h)) / h;
        case 2: return logf(gw / w);
        default: return logf(gh / h);
    }
}
__device__ __forceinline__ float smooth_l1(float d) { const float a = fabsf(d); return a < 1.0f ? 0.5f * d * d : a - 0.5f; }
__device__ __forceinline__ float smooth_l1_grad(float d) { return fabsf(d) < 1.0f ? d : (d > 0.f ? 1.0f : -1.0f); }

template <typename T>
__device__ __forceinline__ float mean_logit(const HeadParams& p, int n, int c) {
    float s = 0.f;
    for (int t = 0; t < p.Tl; ++t) s += ld<T>(p.logits, (size_t)(n * p.Tl + t) * p.lcs + c);
    return s / (float)p.Tl;
}
// the three regression predictions of tube n, coordinate k, and their targets / masks: j = 0 centre, 1 first, 2 last
template <typename T>
__device__ __forceinline__ void reg_terms(const HeadParams& p, int n, int k, float (&pred)[3], float (&tgt)[3], float (&msk)[3]) {
    const int fr[3] = {p.c_mid, p.c_first, p.c_last};
    const int tg[3] = {1, 0, 2};                                          // targets[:, 1] centre, [:, 0] first, [:, -1] last
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const size_t row = (size_t)(n * p.Tl + fr[j]);
        float v = ld<T>(p.reg, row * p.rcs + k);
        if (j) v = v + ld<T>(p.reg, row * p.rcs + 4 * j + k);              // first: columns 4..7, last: 8..11 (two_branch.py:265-270)
        pred[j] = v;
        const float* tr = p.targets + (size_t)n * p.tstride + (size_t)tg[j] * (p.tstride / 3);
        tgt[j] = encode_k(tr, p.tubes + ((size_t)n * p.Tl + fr[j]) * 5 + 1, k);
        msk[j] = tr[5];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void head_outputs_kernel(HeadParams p) {
    __shared__ float red[256];
    const int tid = threadIdx.x;
    const bool train = p.targets != nullptr;
    float pos = 0.f;
    if (train) {
        float m = 0.f;
        for (int n = tid; n < p.N; n += 256) m += p.targets[(size_t)n * p.tstride + (p.tstride / 3) + 4];
        pos = block_sum(m, red) > 0.f ? 1.f : 0.f;                          // `if mask.sum():` (two_branch.py:289)
    }
    // (inference launches several workgroups over the element-wise parts; training -- whose losses are block sums -- exactly one)
    const int gtid = blockIdx.x * 256 + tid, gstep = gridDim.x * 256;
    for (int i = gtid; i < p.N * p.NC; i += gstep) {
        const int n = i / p.NC, c = i % p.NC;
        const float x = mean_logit<T>(p, n, c);
        p.prob[i] = 1.0f / (1.0f + expf(-x));
        if (train) {
            const float* ct = p.targets + (size_t)n * p.tstride + (p.tstride / 3);
            const float t = ct[6 + c] * ct[4];
            p.loss_cls[i] = pos * (fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x))));
        }
    }
    if (p.reg) {
        for (int i = gtid; i < p.N * p.Tl * 4; i += gstep) {
            const int k = i & 3, row = i >> 2;
            p.local_loc[i] = ld<T>(p.reg, (size_t)row * p.rcs + k);
        }
        for (int i = gtid; i < p.N * p.T * 4; i += gstep) {
            const int k = i & 3, t = (i >> 2) % p.T, n = (i >> 2) / p.T;
            const size_t r1 = (size_t)(n * p.Tl + p.lo + t), r2 = (size_t)(n * p.Tl + p.lo2 + t);
            p.first_loc[i] = ld<T>(p.reg, r1 * p.rcs + k) + ld<T>(p.reg, r1 * p.rcs + 4 + k);
            p.last_loc[i] = ld<T>(p.reg, r2 * p.rcs + k) + ld<T>(p.reg, r2 * p.rcs + 8 + k);
        }
        if (train) {
            float sl = 0.f, sm = 0.f, snl = 0.f, snm = 0.f;
            for (int i = tid; i < p.N * 4; i += 256) {
                float pred[3], tgt[3], msk[3];
                reg_terms<T>(p, i >> 2, i & 3, pred, tgt, msk);
                sl += smooth_l1(pred[0] - tgt[0]) * msk[0]; sm += msk[0];
                snl += smooth_l1(pred[1] - tgt[1]) * msk[1] + smooth_l1(pred[2] - tgt[2]) * msk[2]; snm += msk[1] + msk[2];
            }
            const float L = block_sum(sl, red), M = block_sum(sm, red), NL = block_sum(snl, red), NM = block_sum(snm, red);
            if (tid == 0) {
                p.loss_loc[0] = M > 0.f ? L / M : 0.f;
                p.loss_nbr[0] = NM > 0.f ? NL / NM : 0.f;
            }
        }
    }
    if (!train && gtid < 3) {                                                // inference: three separate zero losses (two_branch.py:276-278)
        if (tid == 0) p.loss_cls[0] = 0.f;
        if (tid == 1) p.loss_loc[0] = 0.f;
        if (tid == 2) p.loss_nbr[0] = 0.f;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void head_outputs_bwd_kernel(HeadParams p) {
    __shared__ float red[256];
    const int tid = threadIdx.x;
    float m = 0.f, sm = 0.f, snm = 0.f;
    for (int n = tid; n < p.N; n += 256) {
        const float* t3 = p.targets + (size_t)n * p.tstride;
        const int s = p.tstride / 3;
        m += t3[s + 4]; sm += 4.f * t3[s + 5]; snm += 4.f * (t3[5] + t3[2 * s + 5]);
    }
    const float pos = block_sum(m, red) > 0.f ? 1.f : 0.f;
    const float M = block_sum(sm, red), NM = block_sum(snm, red);
    const float gl = (p.g_loc && M > 0.f) ? p.g_loc[0] / M : 0.f;
    const float gn = (p.g_nbr && NM > 0.f) ? p.g_nbr[0] / NM : 0.f;
    // d loss_cls[n][c] / d logits[n][t][c] = (sigmoid(x) - t) * pos / Tl
    for (int i = tid; i < p.N * p.NC; i += 256) {
        const int n = i / p.NC, c = i % p.NC;
        float g = 0.f;
        if (p.g_cls) {
            const float x = mean_logit<T>(p, n, c);
            const float* ct = p.targets + (size_t)n * p.tstride + (p.tstride / 3);
            g = p.g_cls[i] * pos * (1.0f / (1.0f + expf(-x)) - ct[6 + c] * ct[4]) / (float)p.Tl;
        }
        for (int t = 0; t < p.Tl; ++t) st<T>(p.g_logits, (size_t)(n * p.Tl + t) * p.NC + c, g);
    }
    if (p.g_reg) {
        for (int i = tid; i < p.N * p.Tl * 12; i += 256) {
            const int j = i % 12, t = (i / 12) % p.Tl, n = i / (12 * p.Tl);
            const int k = j & 3, blk = j >> 2;                               // blk 0: local_reg, 1: neighbor_reg1 (first), 2: neighbor_reg2 (last)
            float v = 0.f;
            if (t == p.c_mid || t == p.c_first || t == p.c_last) {
                float pred[3], tgt[3], msk[3];
                reg_terms<T>(p, n, k, pred, tgt, msk);
                const float gc = gl * msk[0] * smooth_l1_grad(pred[0] - tgt[0]);
                const float gf = gn * msk[1] * smooth_l1_grad(pred[1] - tgt[1]);
                const float gla = gn * msk[2] * smooth_l1_grad(pred[2] - tgt[2]);
                if (blk == 0) v = (t == p.c_mid ? gc : 0.f) + (t == p.c_first ? gf : 0.f) + (t == p.c_last ? gla : 0.f);
                else if (blk == 1) v = t == p.c_first ? gf : 0.f;
                else v = t == p.c_last ? gla : 0.f;
            }
            st<T>(p.g_reg, (size_t)i, v);
        }
    }
}

}  // namespace step

using namespace step;

extern "C" {

static int head_fill(HeadParams& p, int dtype, const void* logits, int lcs, const void* reg, int rcs, int N, int Tl, int T, int NC,
                     const float* tubes, const float* targets) {
    if (N < 0 || Tl <= 0 || T <= 0 || NC <= 0 || (Tl % T) != 0 || lcs < NC || (reg && rcs < 12)) return STEP_E_SHAPE;
    if (dtype != STEP_F32 && dtype != STEP_BF16 && dtype != STEP_F16) return STEP_E_DTYPE;
    if ((long long)N * Tl * (NC > 12 ? NC : 12) > 0x7fffffffLL) return STEP_E_UNSUPPORTED;
    const int chunks = Tl / T, half = T / 2;
    p.logits = logits; p.lcs = lcs; p.reg = reg; p.rcs = rcs;
    p.N = N; p.Tl = Tl; p.T = T; p.NC = NC;
    p.c_first = half; p.c_mid = (chunks / 2) * T + half; p.c_last = (chunks - 1) * T + half;      // chunk_idx[0], [chunks / 2], [-1]
    p.lo = 0; p.lo2 = (chunks - 1) * T;
    p.tubes = tubes; p.targets = targets; p.tstride = 3 * (6 + NC);
    return STEP_OK;
}

int step_head_outputs(int dtype, const void* logits, int logits_stride, const void* reg, int reg_stride, int N, int Tl, int T, int NC,
                      const float* tubes, const float* targets, float* prob, float* local_loc, float* first_loc, float* last_loc,
                      float* loss_cls, float* loss_loc, float* loss_nbr, step_stream_t stream) {
    HeadParams p = {};
    const int rc = head_fill(p, dtype, logits, logits_stride, reg, reg_stride, N, Tl, T, NC, tubes, targets);
    if (rc) return rc;
    if (!loss_cls || !loss_loc || !loss_nbr) return STEP_E_NULL;
    if (N > 0 && (!logits || !prob || (reg && (!local_loc || !first_loc || !last_loc)) || (targets && reg && !tubes))) return STEP_E_NULL;
    p.prob = prob; p.local_loc = local_loc; p.first_loc = first_loc; p.last_loc = last_loc;
    p.loss_cls = loss_cls; p.loss_loc = loss_loc; p.loss_nbr = loss_nbr;
    const long long work = (long long)N * (NC > Tl * 4 ? NC : Tl * 4);
    const dim3 grid(targets ? 1u : (unsigned)(work <= 256 ? 1 : (work + 255) / 256 > 64 ? 64 : (work + 255) / 256));
    switch (dtype) {
        case STEP_F32: STEP_LAUNCH((head_outputs_kernel<float>), grid, dim3(256), stream, p); break;
        case STEP_BF16: STEP_LAUNCH((head_outputs_kernel<bf16_t>), grid, dim3(256), stream, p); break;
        default: STEP_LAUNCH((head_outputs_kernel<f16_t>), grid, dim3(256), stream, p); break;
    }
    return STEP_LAUNCH_CHECK();
}

int step_head_outputs_backward(int dtype, const void* logits, int logits_stride, const void* reg, int reg_stride, int N, int Tl, int T, int NC,
                               const float* tubes, const float* targets, const float* g_loss_cls, const float* g_loss_loc,
                               const float* g_loss_nbr, void* g_logits, void* g_reg, step_stream_t stream) {
    HeadParams p = {};
    const int rc = head_fill(p, dtype, logits, logits_stride, reg, reg_stride, N, Tl, T, NC, tubes, targets);
    if (rc) return rc;
    if (N == 0) return STEP_OK;
    if (!logits || !targets || !g_logits || (reg && (!tubes || !g_reg))) return STEP_E_NULL;
    p.g_cls = g_loss_cls; p.g_loc = g_loss_loc; p.g_nbr = g_loss_nbr; p.g_logits = g_logits; p.g_reg = reg ? g_reg : nullptr;
    switch (dtype) {
        case STEP_F32: STEP_LAUNCH((head_outputs_bwd_kernel<float>), dim3(1), dim3(256), stream, p); break;
        case STEP_BF16: STEP_LAUNCH((head_outputs_bwd_kernel<bf16_t>), dim3(1), dim3(256), stream, p); break;
        default: STEP_LAUNCH((head_outputs_bwd_kernel<f16_t>), dim3(1), dim3(256), stream, p); break;
    }
    return STEP_LAUNCH_CHECK();
}

}  // extern "C"
