"""Backend-agnostic parity cases for the C ABI (include/step_amd.h): every case runs the kernels
through `bk` (tests/backends.py: host interpreter or the real gfx950 library) and checks the result
against the oracle (oracle/) or the golden vectors (tests/golden)."""
import ctypes
import os

import numpy as np
import torch
import torch.nn.functional as F

import oracle
from oracle import i3d_ref as R
from step_amd import _capi

F32, BF16, F16 = _capi.F32, _capi.BF16, _capi.F16
NCHW, NHWC = _capi.NCHW, _capi.NHWC
NP_DT = {F32: np.float32, BF16: np.uint16, F16: np.float16}


# ------------------------------------------------------------------ dtype helpers
def to_bf16_bits(x):
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def from_bf16_bits(b):
    return (b.astype(np.uint32) << 16).view(np.float32)


def encode(x, dt):
    x = np.ascontiguousarray(x, np.float32)
    return x if dt == F32 else (to_bf16_bits(x) if dt == BF16 else x.astype(np.float16))


def decode(a, dt):
    return a if dt == F32 else (from_bf16_bits(a) if dt == BF16 else a.astype(np.float32))


def quantize(x, dt):
    return decode(encode(x, dt), dt)


def tol(dt):
    # fp32: summation-order noise only.  16-bit: one rounding of the stored output (inputs are
    # quantised identically on both sides, accumulation is fp32 on both sides).
    return {F32: 2e-5, BF16: 2 ** -7, F16: 2 ** -9}[dt]


def nhwc(x):
    return np.ascontiguousarray(np.transpose(x, (0, 2, 3, 1)))


def nchw(x):
    return np.ascontiguousarray(np.transpose(x, (0, 3, 1, 2)))


def cl(x):   # NCDHW -> NDHWC
    return np.ascontiguousarray(np.transpose(x, (0, 2, 3, 4, 1)))


def uncl(x):
    return np.ascontiguousarray(np.transpose(x, (0, 4, 1, 2, 3)))


# ------------------------------------------------------------------ op runners
def run_roi_align(bk, feat_nchw, rois, pooled, scale, sr, layout, dt=F32):
    B, C, H, W = feat_nchw.shape
    K = rois.shape[0]
    ph, pw = pooled
    f = bk.dev(encode(feat_nchw if layout == NCHW else nhwc(feat_nchw), dt))
    r = bk.dev(np.ascontiguousarray(rois, np.float32))
    out = bk.dev(np.zeros((K, C, ph, pw) if layout == NCHW else (K, ph, pw, C), NP_DT[dt]))
    rc = bk.lib.step_roi_align_forward(f.ptr, dt, layout, r.ptr, K, B, C, H, W, ph, pw, scale, sr, out.ptr, bk.stream)
    assert rc == 0, rc
    o = decode(out.get(), dt)
    return o if layout == NCHW else nchw(o)


def run_nms(bk, boxes, scores, counts, thr):
    G, kmax = scores.shape
    b, s, c = bk.dev(boxes), bk.dev(scores), bk.dev(counts)
    keep = bk.dev(np.full((G, kmax), 9, np.uint8))
    nb = bk.lib.step_nms_scratch_bytes(G, kmax)
    scratch = bk.dev(np.zeros(max(nb, 1), np.uint8))
    rc = bk.lib.step_nms_batched(b.ptr, s.ptr, c.ptr, G, kmax, thr, keep.ptr, scratch.ptr if nb else None, bk.stream)
    assert rc == 0, rc
    return keep.get()


def pack_weight(bk, w, dt, perm=None):
    Cout, Cin, kd, kh, kw = w.shape
    n = bk.lib.step_conv_packed_elems(Cout, Cin, kd, kh, kw)
    out = bk.dev(np.zeros(n, NP_DT[dt]))
    wd = bk.dev(np.ascontiguousarray(w, np.float32))
    p = bk.dev(None if perm is None else np.ascontiguousarray(perm, np.int32))
    assert bk.lib.step_conv_pack_weight(wd.ptr, Cout, Cin, kd, kh, kw, dt, p.ptr, out.ptr, bk.stream) == 0
    return out


def run_conv(bk, x, w, scale, shift, dt, relu=True, res=None, x_pad=(0, 0), y_pad=(0, 0), use_ws=False):
    """x NCDHW fp32, w torch layout; x_pad/y_pad = extra channels (before, after) in the buffers.
    use_ws: call step_conv_forward_ws with the scratch buffer step_conv_workspace_bytes asks for."""
    N, Cin, D, H, W = x.shape
    Cout = w.shape[0]
    xc = cl(x)
    xb = np.zeros(xc.shape[:-1] + (x_pad[0] + Cin + x_pad[1],), np.float32)
    xb[..., x_pad[0]:x_pad[0] + Cin] = xc
    xb[..., :x_pad[0]] = 77.0           # must never be read
    xb[..., x_pad[0] + Cin:] = -55.0
    xe = bk.dev(encode(xb, dt))
    yb = bk.dev(np.zeros((N, D, H, W, y_pad[0] + Cout + y_pad[1]), NP_DT[dt]))
    wp = pack_weight(bk, w, dt)
    d = _capi.ConvDesc(dtype=dt, N=N, D=D, H=H, W=W, Cin=Cin, Cout=Cout, kd=w.shape[2], kh=w.shape[3], kw=w.shape[4],
                       x_cstride=xb.shape[-1], x_coff=x_pad[0], y_cstride=y_pad[0] + Cout + y_pad[1], y_coff=y_pad[0],
                       res_cstride=Cout, res_coff=0, relu=int(relu), split=0, y2_cstride=0, y2_coff=0)
    re = bk.dev(None if res is None else encode(cl(res), dt))
    sc = bk.dev(None if scale is None else np.ascontiguousarray(scale, np.float32))
    sh = bk.dev(None if shift is None else np.ascontiguousarray(shift, np.float32))
    if use_ws:
        nb = bk.lib.step_conv_workspace_bytes(ctypes.byref(d))
        run_conv.last_ws_bytes = nb
        ws = bk.dev(np.full(max(nb, 16) // 4, np.nan, np.float32))      # scratch needs no initialisation: poison it
        rc = bk.lib.step_conv_forward_ws(ctypes.byref(d), xe.ptr, wp.ptr, sc.ptr, sh.ptr, re.ptr, yb.ptr, None, ws.ptr, nb, bk.stream)
    else:
        rc = bk.lib.step_conv_forward(ctypes.byref(d), xe.ptr, wp.ptr, sc.ptr, sh.ptr, re.ptr, yb.ptr, None, bk.stream)
    assert rc == 0, rc
    y = decode(yb.get(), dt)
    assert not y[..., :y_pad[0]].any() and not y[..., y_pad[0] + Cout:].any()
    return uncl(y[..., y_pad[0]:y_pad[0] + Cout])


def ref_conv(x, w, scale, shift, dt, relu=True, res=None):
    xq = torch.from_numpy(quantize(x, dt))
    wq = torch.from_numpy(quantize(w, dt))
    y = F.conv3d(xq, wq, padding=tuple(k // 2 for k in w.shape[2:]))
    if scale is not None:
        y = y * torch.from_numpy(scale).view(1, -1, 1, 1, 1)
    if shift is not None:
        y = y + torch.from_numpy(shift).view(1, -1, 1, 1, 1)
    if res is not None:
        y = y + torch.from_numpy(quantize(res, dt))
    return (F.relu(y) if relu else y).numpy()


def run_stem(bk, x_ntchw, w, scale, shift, dt):
    N, T, _, H, W = x_ntchw.shape
    Cout = w.shape[0]
    wp = bk.dev(np.zeros(bk.lib.step_stem_packed_elems(Cout), NP_DT[dt]))
    wd = bk.dev(np.ascontiguousarray(w, np.float32))
    assert bk.lib.step_stem_pack_weight(wd.ptr, Cout, dt, wp.ptr, bk.stream) == 0
    To, Ho, Wo = (T - 2) // 2 + 1, (H - 2) // 2 + 1, (W - 2) // 2 + 1
    y = bk.dev(np.zeros((N, To, Ho, Wo, Cout), NP_DT[dt]))
    xe, sc, sh = bk.dev(encode(x_ntchw, dt)), bk.dev(scale), bk.dev(shift)
    rc = bk.lib.step_stem_forward(dt, xe.ptr, N, T, H, W, wp.ptr, sc.ptr, sh.ptr, 1, Cout, y.ptr, Cout, 0, bk.stream)
    assert rc == 0, rc
    return uncl(decode(y.get(), dt))


def ref_stem(x, w, scale, shift, dt):
    xq = torch.from_numpy(quantize(x, dt)).permute(0, 2, 1, 3, 4)
    y = F.conv3d(F.pad(xq, (2, 3, 2, 3, 2, 3)), torch.from_numpy(quantize(w, dt)), stride=2)
    return F.relu(y * torch.from_numpy(scale).view(1, -1, 1, 1, 1) + torch.from_numpy(shift).view(1, -1, 1, 1, 1)).numpy()


# ------------------------------------------------------------------ cases (bk, golden)
def case_roi_align_forward_golden_bit_exact(bk, golden):
    g = golden("roi_nms_golden")
    feat = R.fill_tensor("golden.roi.feat", (3, 6, 25, 25), "image").numpy()   # C=6: scalar-lane path
    for layout in (NCHW, NHWC):
        for tag, pooled, sr in (("p7s0", (7, 7), 0), ("p7s2", (7, 7), 2), ("p3x5s0", (3, 5), 0)):
            out = run_roi_align(bk, feat, g["align_rois"], pooled, 1 / 16., sr, layout)
            assert np.array_equal(out, g["align_out_" + tag]), (layout, tag)


def case_roi_align_forward_vector_path_and_tubes(bk, golden):
    g = golden("roi_nms_golden")
    conv = R.fill_tensor("golden.roi.conv", (2, 3, 16, 25, 25), "feat").numpy().reshape(6, 16, 25, 25)
    out = run_roi_align(bk, conv, g["tube_rois"].reshape(-1, 5), (7, 7), 1 / 16., 0, NHWC)   # C=16: 16-byte lanes
    assert np.array_equal(out, g["tube_out"])


def case_roi_align_forward_16bit(bk, golden):
    rs = np.random.RandomState(1)
    x = rs.randn(2, 16, 13, 11).astype(np.float32)
    rois = np.array([[0, 3, 5, 150, 120], [1, 20, 30, 60.5, 99.25]], np.float32)
    for dt in (BF16, F16):
        ref = oracle.roi_align_forward(quantize(x, dt), rois, (7, 7), 1 / 16., 0)
        for layout in (NCHW, NHWC):
            out = run_roi_align(bk, x, rois, (7, 7), 1 / 16., 0, layout, dt)
            assert np.abs(out - ref).max() <= tol(dt) * np.abs(ref).max()


def case_roi_align_backward(bk, golden):
    """step_roi_align_backward against the oracle's restatement of ROIAlign_cpu.cpp's backward: the default fixed-order gather
    (bit-reproducible, every cell written -- frames without a roi come back zero) and the reference's atomics scatter, chosen PER
    CALL (the `mode` argument; an unknown mode is refused); rois that hang over every border, a degenerate one, adaptive and fixed sampling, vector and scalar
    channel counts."""
    rs = np.random.RandomState(2)
    few = np.array([[0, 0, 0, 190, 140], [1, 33.3, 20.1, 120.7, 100.2], [1, -20, 100, 90, 250], [0, 50, 50, 50.5, 50.5]], np.float32)
    many = np.concatenate([rs.randint(0, 2, (40, 1)).astype(np.float32), rs.uniform(-30, 150, (40, 2)).astype(np.float32),
                           rs.uniform(60, 260, (40, 2)).astype(np.float32)], 1)
    many[::7, 3:] = many[::7, 1:3] - 5.0                      # malformed (x2 < x1): forced to 1 x 1
    for gather in (1, 0):
        mode = _capi.ROI_BWD_GATHER if gather else _capi.ROI_BWD_ATOMIC
        if True:
            for rois, B, C, H, W in ((few, 2, 8, 9, 12), (many, 3, 8, 9, 12), (many, 2, 5, 11, 7)):
                K = rois.shape[0]
                for layout in (NCHW, NHWC):
                    for sr in (0, 2):
                        g = rs.randn(K, C, 7, 7).astype(np.float32)
                        ref = oracle.roi_align_backward(g, rois, (7, 7), 1 / 16., sr, (B, C, H, W))
                        gg, r = bk.dev(g if layout == NCHW else nhwc(g)), bk.dev(rois)
                        outs = []
                        for _ in range(2):
                            gi = bk.dev(np.full((B, C, H, W) if layout == NCHW else (B, H, W, C), 7.0, np.float32))   # the op owns every cell
                            rc = bk.lib.step_roi_align_backward(gg.ptr, layout, r.ptr, K, B, C, H, W, 7, 7, 1 / 16., sr, mode, gi.ptr, bk.stream)
                            assert rc == 0
                            outs.append(gi.get() if layout == NCHW else nchw(gi.get()))
                        assert np.abs(outs[0] - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max()), (gather, layout, sr, B, C)   # summation order differs
                        if B == 3:
                            assert not outs[0][2].any()
                        if gather:
                            assert np.array_equal(outs[0], outs[1])
    # the gather's per-cell roi list: more rois than one 64-header scan round (order must stay ascending across rounds), and more rois
    # on one cell than the list holds (520 whole-frame rois on frame 0 > 512: that cell walks every header instead)
    spread = np.concatenate([rs.randint(0, 3, (150, 1)).astype(np.float32), rs.uniform(-10, 100, (150, 2)).astype(np.float32),
                             rs.uniform(60, 200, (150, 2)).astype(np.float32)], 1)
    crowd = np.tile(np.array([[0, 0, 0, 190, 140]], np.float32), (520, 1))
    crowd[::3, 1:] += rs.uniform(-3, 3, (len(crowd[::3]), 4)).astype(np.float32)
    for rois, B, C, H, W in ((spread, 3, 8, 9, 12), (crowd, 1, 4, 9, 12)):
        K = rois.shape[0]
        g = rs.randn(K, C, 7, 7).astype(np.float32)
        ref = oracle.roi_align_backward(g, rois, (7, 7), 1 / 16., 0, (B, C, H, W))
        gg, r = bk.dev(nhwc(g)), bk.dev(rois)
        outs = []
        for _ in range(2):
            gi = bk.dev(np.full((B, H, W, C), 7.0, np.float32))
            assert bk.lib.step_roi_align_backward(gg.ptr, NHWC, r.ptr, K, B, C, H, W, 7, 7, 1 / 16., 0, _capi.ROI_BWD_GATHER, gi.ptr, bk.stream) == 0
            outs.append(nchw(gi.get()))
        assert np.abs(outs[0] - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max()), K
        assert np.array_equal(outs[0], outs[1])
    gi = bk.dev(np.full((2, 9, 12, 8), 7.0, np.float32))
    assert bk.lib.step_roi_align_backward(None, NHWC, None, 0, 2, 8, 9, 12, 7, 7, 1 / 16., 2, _capi.ROI_BWD_ATOMIC, gi.ptr, bk.stream) == 0 and not gi.get().any()
    assert bk.lib.step_roi_align_backward(None, NHWC, None, 0, 2, 8, 9, 12, 7, 7, 1 / 16., 2, 2, gi.ptr, bk.stream) == -4       # unknown mode


def case_roi_pool_forward_backward(bk, golden):
    rs = np.random.RandomState(3)
    B, C, H, W = 2, 5, 10, 13
    x = rs.randn(B, C, H, W).astype(np.float32)
    rois = np.array([[0, 0, 0, 200, 150], [1, 17, 9, 88, 140], [0, 300, 300, 400, 400], [1, 40, 40, 41, 41], [0, -30, -10, 50, 60]], np.float32)
    K = rois.shape[0]
    ref, refarg = oracle.roi_pool_forward(x, rois, (7, 7), 1 / 16.)
    for layout in (NCHW, NHWC):
        xx, r = bk.dev(x if layout == NCHW else nhwc(x)), bk.dev(rois)
        oshape = (K, C, 7, 7) if layout == NCHW else (K, 7, 7, C)
        out, arg = bk.dev(np.zeros(oshape, np.float32)), bk.dev(np.zeros(oshape, np.int32))
        assert bk.lib.step_roi_pool_forward(xx.ptr, F32, layout, r.ptr, K, B, C, H, W, 7, 7, 1 / 16., out.ptr, arg.ptr, bk.stream) == 0
        o, a = (out.get(), arg.get()) if layout == NCHW else (nchw(out.get()), nchw(arg.get()))
        assert np.array_equal(o, ref) and np.array_equal(a, refarg)
        g = rs.randn(K, C, 7, 7).astype(np.float32)
        refg = oracle.roi_pool_backward(g, refarg, rois, (7, 7), x.shape)
        gg = bk.dev(g if layout == NCHW else nhwc(g))
        gi = bk.dev(np.full(xx.get().shape, 3.0, np.float32))
        assert bk.lib.step_roi_pool_backward(gg.ptr, arg.ptr, layout, r.ptr, K, B, C, H, W, 7, 7, gi.ptr, bk.stream) == 0
        got = gi.get() if layout == NCHW else nchw(gi.get())
        assert np.abs(got - refg).max() <= 1e-5


def case_roi_pool_hand_derived_vectors(bk, golden):
    """step_roi_pool_forward / _backward against the 12 cases derived by hand from the reference's CUDA kernel
    (tests/golden/make_roipool_hand_vectors.py, ROIPool_cuda.cu:40-132): value, argmax and gradient scatter, both layouts, bit for bit;
    the 16-bit storage types give the same argmax and the same (exactly representable) values."""
    import json
    vec = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "roipool_hand_vectors.json")))
    for v in vec:
        x = np.array(v["feature"], np.float32)
        rois = np.array(v["rois"], np.float32)
        B, C, H, W = x.shape
        K = rois.shape[0]
        PH, PW = v["pooled"]
        want, wantarg = np.array(v["out"], np.float32), np.array(v["argmax"], np.int32)
        for layout in (NCHW, NHWC):
            for dt in (F32, BF16, F16):
                xx, r = bk.dev(encode(x if layout == NCHW else nhwc(x), dt)), bk.dev(rois)
                oshape = (K, C, PH, PW) if layout == NCHW else (K, PH, PW, C)
                out, arg = bk.dev(np.zeros(oshape, NP_DT[dt])), bk.dev(np.full(oshape, 7, np.int32))
                assert bk.lib.step_roi_pool_forward(xx.ptr, dt, layout, r.ptr, K, B, C, H, W, PH, PW, v["spatial_scale"], out.ptr, arg.ptr,
                                                    bk.stream) == 0
                o, a = decode(out.get(), dt), arg.get()
                if layout == NHWC:
                    o, a = nchw(o), nchw(a)
                assert np.array_equal(a, wantarg), (v["name"], layout, dt, v["decided_by"])
                assert np.array_equal(o, want), (v["name"], layout, dt, v["decided_by"])          # (integers <= 178: exact in bf16 / fp16)
            g = np.array(v["grad_out"], np.float32)
            gg = bk.dev(g if layout == NCHW else nhwc(g))
            argd = bk.dev(wantarg if layout == NCHW else nhwc(wantarg))
            gi = bk.dev(np.full((B, C, H, W) if layout == NCHW else (B, H, W, C), 3.0, np.float32))           # the op zeroes it
            assert bk.lib.step_roi_pool_backward(gg.ptr, argd.ptr, layout, r.ptr, K, B, C, H, W, PH, PW, gi.ptr, bk.stream) == 0
            got = gi.get() if layout == NCHW else nchw(gi.get())
            assert np.array_equal(got, np.array(v["grad_in"], np.float32)), (v["name"], layout)


def case_nms_fp64(bk, golden):
    """step_nms_batched_f64 (double boxes / scores, as the reference dispatches them) against the fp64 restatement -- itself bit-identical
    to the reference's nms_cpu_kernel<double> (tests/test_oracle_vs_reference.py) -- on boxes whose IoU sits so close to the threshold
    that the fp32 arithmetic decides differently; wave (k <= 64) and block (k > 64) kernels."""
    rs = np.random.RandomState(77)
    flips = 0
    for k in (34, 64, 150):
        G = 6
        xy = rs.uniform(0, 300, (G, k, 2))
        boxes = np.concatenate([xy, xy + rs.uniform(20, 120, (G, k, 2))], 2)
        # plant pairs whose IoU is the threshold +- 1e-9: box j = box i shifted so that inter / union = 0.4 to within double rounding
        for g in range(G):
            for i in range(0, k - 1, 7):
                x1, y1, x2, y2 = boxes[g, i]
                w, h = x2 - x1 + 1, y2 - y1 + 1
                # overlap o along x with equal sizes: iou = o h / (2 w h - o h) = 0.4 -> o = 0.8 w / 1.4
                o = 0.8 * w / 1.4 + rs.choice([-1e-9, 1e-9])
                boxes[g, i + 1] = [x1 + (w - o), y1, x2 + (w - o), y2]
        scores = rs.permutation(G * k).reshape(G, k).astype(np.float64) / (G * k)
        counts = np.array([k, k - 3, 1, 0, k, k // 2], np.int32)
        want = np.zeros((G, k), np.uint8)
        for g in range(G):
            n = counts[g]
            want[g, oracle.nms_f64(boxes[g, :n], scores[g, :n], 0.4)] = 1
            if n and not np.array_equal(oracle.nms_f64(boxes[g, :n], scores[g, :n], 0.4), oracle.nms(boxes[g, :n], scores[g, :n], 0.4)):
                flips += 1
        b, s_, c = bk.dev(boxes), bk.dev(scores), bk.dev(counts)
        keep = bk.dev(np.full((G, k), 9, np.uint8))
        nb = bk.lib.step_nms_scratch_bytes(G, k)
        scratch = bk.dev(np.zeros(max(nb, 16), np.uint8))
        assert bk.lib.step_nms_batched_f64(b.ptr, s_.ptr, c.ptr, G, k, 0.4, keep.ptr, scratch.ptr, bk.stream) == 0
        assert np.array_equal(keep.get(), want), k
    assert flips >= 1, "the planted near-threshold pairs should make fp32 and fp64 disagree somewhere"


def case_roi_empty_and_bad_args(bk, golden):
    L = bk.lib
    x = bk.dev(np.zeros((1, 4, 4, 8), np.float32))
    assert L.step_roi_align_forward(x.ptr, 0, 1, None, 0, 1, 8, 4, 4, 7, 7, 1.0, 0, None, bk.stream) == 0      # K = 0
    assert L.step_roi_align_forward(x.ptr, 9, 1, x.ptr, 1, 1, 8, 4, 4, 7, 7, 1.0, 0, x.ptr, bk.stream) == -1   # dtype
    assert L.step_roi_align_forward(None, 0, 1, x.ptr, 1, 1, 8, 4, 4, 7, 7, 1.0, 0, x.ptr, bk.stream) == -3    # null
    assert L.step_roi_align_forward(x.ptr, 0, 5, x.ptr, 1, 1, 8, 4, 4, 7, 7, 1.0, 0, x.ptr, bk.stream) == -4   # layout
    assert L.step_nms_batched(None, None, None, 0, 34, 0.4, None, None, bk.stream) == 0                          # G = 0
    assert L.step_abi_version() == _capi.ABI_VERSION and b"gfx950" in L.step_version()


def case_nms_golden_bit_exact(bk, golden):
    g = golden("roi_nms_golden")
    for i in range(int(g["nms_count"])):
        b, s, thr = g["nms%d_boxes" % i], g["nms%d_scores" % i], float(g["nms%d_thr" % i])
        keep = run_nms(bk, b[None].copy(), s[None].copy(), np.array([b.shape[0]], np.int32), thr)
        assert np.array_equal(np.nonzero(keep[0])[0], g["nms%d_keep" % i]), i


def case_nms_batched_groups_and_ties(bk, golden):
    for kmax in (11, 34, 64, 109):
        rs = np.random.RandomState(kmax)
        G = 9
        boxes = np.zeros((G, kmax, 4), np.float32)
        scores = np.zeros((G, kmax), np.float32)
        counts = rs.randint(0, kmax + 1, G).astype(np.int32)
        counts[0], counts[1] = 0, kmax
        for gi in range(G):
            xy = rs.uniform(0, 300, (kmax, 2))
            wh = rs.uniform(10, 150, (kmax, 2))
            boxes[gi] = np.concatenate([xy, xy + wh], 1)
            scores[gi] = rs.randint(0, 6, kmax) / 6.0          # many exact ties -> lower index first
        keep = run_nms(bk, boxes, scores, counts, 0.4)
        assert np.array_equal(keep, oracle.nms_batched(boxes, scores, counts, 0.4)), kmax


def case_nms_nan_scores(bk, golden):
    """NaN scores rank first (torch's sort order, which the reference inherits); ranks stay a permutation, so neither the
    wave kernel's rank -> lane lookup nor the block kernel's order table is undefined."""
    for kmax in (12, 34, 80):
        rs = np.random.RandomState(7 + kmax)
        G = 5
        boxes = np.zeros((G, kmax, 4), np.float32)
        scores = np.zeros((G, kmax), np.float32)
        for gi in range(G):
            xy = rs.uniform(0, 200, (kmax, 2))
            boxes[gi] = np.concatenate([xy, xy + rs.uniform(20, 150, (kmax, 2))], 1)
            scores[gi] = rs.permutation(kmax) / kmax
            scores[gi, rs.choice(kmax, gi, replace=False)] = np.nan      # 0..4 NaNs per group
        counts = np.full(G, kmax, np.int32)
        keep = run_nms(bk, boxes, scores, counts, 0.3)
        assert np.array_equal(keep, oracle.nms_batched(boxes, scores, counts, 0.3)), kmax


POOLS = [((1, 3, 3), (1, 2, 2)), ((3, 3, 3), (2, 2, 2)), ((3, 3, 3), (1, 1, 1)), ((2, 2, 2), (2, 2, 2))]


def case_maxpool_tf(bk, golden):
    L = bk.lib
    rs = np.random.RandomState(5)
    N, C, D, H, W = 2, 16, 5, 9, 7                    # odd sizes: ceil-mode overhang, negative values
    x = rs.randn(N, C, D, H, W).astype(np.float32)
    for k, s in POOLS:
        for dt in (F32, BF16):
            ref = R.maxpool_tf(torch.from_numpy(quantize(x, dt)), k, s).numpy()
            Do, Ho, Wo = (L.step_pool_out_size(a, b, c) for a, b, c in zip((D, H, W), k, s))
            assert ref.shape == (N, C, Do, Ho, Wo)
            xcl = bk.dev(encode(cl(x), dt))
            ycs, yoff = 40, 8                          # write into a channel slice of a wider buffer
            y = bk.dev(np.zeros((N, Do, Ho, Wo, ycs), NP_DT[dt]))
            rc = L.step_maxpool3d_tf(dt, xcl.ptr, N, D, H, W, C, C, 0, k[0], k[1], k[2], s[0], s[1], s[2], y.ptr, ycs, yoff, bk.stream)
            assert rc == 0
            yy = decode(y.get(), dt)
            assert np.array_equal(uncl(yy[..., yoff:yoff + C]), ref), (k, s, dt)
            assert not yy[..., :yoff].any() and not yy[..., yoff + C:].any()


def case_maxpool_s1_long_d_segments(bk, golden):
    """3x3x3 stride-1 pool on a long D axis: the rolling-window kernel splits D into segments with halos."""
    rs = np.random.RandomState(8)
    N, C, D, H, W = 1, 16, 17, 9, 19
    x = rs.randn(N, C, D, H, W).astype(np.float32)
    ref = R.maxpool_tf(torch.from_numpy(x), (3, 3, 3), (1, 1, 1)).numpy()
    xd = bk.dev(cl(x))
    y = bk.dev(np.zeros((N, D, H, W, C), np.float32))
    assert bk.lib.step_maxpool3d_tf(0, xd.ptr, N, D, H, W, C, C, 0, 3, 3, 3, 1, 1, 1, y.ptr, C, 0, bk.stream) == 0
    assert np.array_equal(uncl(y.get()), ref)
    # 16-bit storage (the max is taken in the storage type: exact), ragged channel chunk (40 = 32 + 8), 2x2 tiles
    N, C, D, H, W = 2, 40, 3, 15, 16
    x = rs.randn(N, C, D, H, W).astype(np.float32)
    for dt in (BF16, F16):
        ref = R.maxpool_tf(torch.from_numpy(quantize(x, dt)), (3, 3, 3), (1, 1, 1)).numpy()
        xd = bk.dev(encode(cl(x), dt))
        y = bk.dev(np.zeros((N, D, H, W, 48), NP_DT[dt]))
        assert bk.lib.step_maxpool3d_tf(dt, xd.ptr, N, D, H, W, C, C, 0, 3, 3, 3, 1, 1, 1, y.ptr, 48, 8, bk.stream) == 0
        yy = decode(y.get(), dt)
        assert np.array_equal(uncl(yy[..., 8:]), ref), dt
        assert not yy[..., :8].any()


def case_maxpool_zero_pad_value(bk, golden):
    g = golden("ops_golden")
    x = bk.dev(-np.ones((1, 4, 5, 5, 4), np.float32))
    y = bk.dev(np.zeros((1, 2, 3, 3, 4), np.float32))
    assert bk.lib.step_maxpool3d_tf(0, x.ptr, 1, 4, 5, 5, 4, 4, 0, 3, 3, 3, 2, 2, 2, y.ptr, 4, 0, bk.stream) == 0
    yy = uncl(y.get())
    assert np.array_equal(yy[:, :1], g["pool_allneg"]) and yy.max() == 0.0 and yy.min() == -1.0


def case_maxpool_backward(bk, golden):
    """Pool backward against torch autograd of ConstantPad3d(0) + MaxPool3d(ceil_mode) (i3dpt.py:114-126), incl. ties
    between real zeros and pad zeros (post-ReLU data) and all-negative windows won by the pad."""
    L = bk.lib
    rs = np.random.RandomState(9)
    N, C, D, H, W = 2, 8, 5, 9, 7
    x = rs.randn(N, C, D, H, W).astype(np.float32)
    x[0] = np.maximum(x[0], 0)                          # clip 0: post-ReLU data, many exact zeros (ties)
    for k, s in POOLS:
        xt = torch.from_numpy(x).requires_grad_(True)
        pads = []
        for kk, ss in zip(reversed(k), reversed(s)):
            a = max(kk - ss, 0)
            pads += [a // 2, a - a // 2]
        y = F.max_pool3d(F.pad(xt, pads), k, s, ceil_mode=True)
        gy = rs.randn(*y.shape).astype(np.float32)
        y.backward(torch.from_numpy(gy))
        ref = xt.grad.numpy()
        xd = bk.dev(cl(x))
        gyd = bk.dev(np.ascontiguousarray(cl(gy)))
        gxd = bk.dev(np.full((N, D, H, W, C), 5.0, np.float32))
        assert L.step_maxpool3d_tf_backward(0, xd.ptr, N, D, H, W, C, C, 0, k[0], k[1], k[2], s[0], s[1], s[2], gyd.ptr, gxd.ptr, bk.stream) == 0
        got = uncl(gxd.get())
        # (fp32 atomics: the order in which up to 27 windows' gradients meet differs from run to run -- a few ulps of the sum)
        assert np.allclose(got, ref, rtol=2e-5, atol=2e-5), (k, s, np.abs(got - ref).max())
        # the gather form (arg map + fixed-order gather): fp32 like the reference; bit-reproducible
        Do, Ho, Wo = y.shape[2:]
        arg = bk.dev(np.full(N * Do * Ho * Wo * C, 250, np.uint8))
        g2 = bk.dev(np.full((N, D, H, W, C), 5.0, np.float32))
        assert L.step_maxpool3d_tf_backward_gather(0, xd.ptr, N, D, H, W, C, C, 0, k[0], k[1], k[2], s[0], s[1], s[2], 0, gyd.ptr, 0, g2.ptr,
                                                   arg.ptr, bk.stream) == 0
        first = uncl(g2.get()).copy()
        assert np.allclose(first, ref, rtol=1e-6, atol=1e-6), (k, s, np.abs(first - ref).max())
        assert L.step_maxpool3d_tf_backward_gather(0, xd.ptr, N, D, H, W, C, C, 0, k[0], k[1], k[2], s[0], s[1], s[2], 0, gyd.ptr, 0, g2.ptr,
                                                   arg.ptr, bk.stream) == 0
        assert np.array_equal(uncl(g2.get()), first)
        # 16-bit activations: x, gy and gx in the activation type (and the mixed forms), against torch on the quantized operands
        for dt in (BF16, F16):
            xq = quantize(x, dt)
            xt = torch.from_numpy(xq).requires_grad_(True)
            yq = F.max_pool3d(F.pad(xt, pads), k, s, ceil_mode=True)
            gq = quantize(gy, dt)
            yq.backward(torch.from_numpy(gq))
            refq = xt.grad.numpy()
            xe = bk.dev(encode(cl(xq), dt))
            for gdt, odt in ((dt, dt), (0, dt), (dt, 0)):
                ge = bk.dev(np.ascontiguousarray(cl(gq)) if gdt == 0 else encode(np.ascontiguousarray(cl(gq)), dt))
                go = bk.dev(np.full((N, D, H, W, C), 3, np.float32 if odt == 0 else NP_DT[dt]))
                assert L.step_maxpool3d_tf_backward_gather(dt, xe.ptr, N, D, H, W, C, C, 0, k[0], k[1], k[2], s[0], s[1], s[2], gdt, ge.ptr, odt,
                                                           go.ptr, arg.ptr, bk.stream) == 0
                out = uncl(go.get() if odt == 0 else decode(go.get(), dt))
                want = refq if odt == 0 else quantize(refq, dt)
                assert np.allclose(out, want, rtol=2e-2 if odt else 1e-6, atol=1e-6), (k, s, dt, gdt, odt, np.abs(out - want).max())
    # the one-launch (3,3,3) / (1,1,1) form (16-bit activations): several tiles per map with ragged last ones, a ragged last channel
    # chunk, one- and two-plane samples, windows with no winner (NaN / -inf everywhere) and NaNs beside real values; against torch
    # and against the two-gather form (pool_direct = 1), which adds the same terms in another order
    for (n2, c2, d2, h2, w2) in ((2, 24, 4, 17, 30), (1, 8, 1, 15, 9), (3, 40, 2, 7, 7), (1, 16, 3, 31, 3)):
        xb = rs.randn(n2, c2, d2, h2, w2).astype(np.float32)
        xb[0, :, :, :h2 // 2] = np.maximum(xb[0, :, :, :h2 // 2], 0)              # ties with the pad and with each other
        xb[0, 0, :, :3, :3] = -np.inf
        xb[0, 1, :, :4, :4] = np.nan
        xb[0, 2, 0, 1, 1] = np.nan
        gb = rs.randn(n2, c2, d2, h2, w2).astype(np.float32)
        for dt in (BF16, F16):
            xq, gq = quantize(xb, dt), quantize(gb, dt)
            xt = torch.from_numpy(np.nan_to_num(xq, nan=-np.inf)).requires_grad_(True)       # torch's pool PROPAGATES NaN; the walk `val > max` skips it
            F.max_pool3d(F.pad(xt, [1, 1, 1, 1, 1, 1]), 3, 1).backward(torch.from_numpy(gq))
            refq = xt.grad.numpy()
            xe, ge = bk.dev(encode(cl(xq), dt)), bk.dev(encode(np.ascontiguousarray(cl(gq)), dt))
            outs = []
            for direct in (0, 1):
                _capi.set_option(L, "pool_direct", direct)
                try:
                    go = bk.dev(np.full((n2, d2, h2, w2, c2), 3, np.float32))
                    argb = bk.dev(np.zeros(n2 * d2 * h2 * w2 * c2, np.uint8))
                    assert L.step_maxpool3d_tf_backward_gather(dt, xe.ptr, n2, d2, h2, w2, c2, c2, 0, 3, 3, 3, 1, 1, 1, dt, ge.ptr, 0, go.ptr, argb.ptr, bk.stream) == 0
                    outs.append(uncl(go.get()).copy())
                finally:
                    _capi.set_option(L, "pool_direct", 0)
            assert np.allclose(outs[0], outs[1], rtol=1e-6, atol=1e-6), ((n2, c2, d2, h2, w2), dt, np.abs(outs[0] - outs[1]).max())
            # (channels 0 and 1 hold windows with no winner: torch hands those to the window's first element, both forms here to nobody)
            assert np.allclose(outs[0][:, 2:], refq[:, 2:], rtol=1e-6, atol=1e-6), ((n2, c2, d2, h2, w2), dt, np.abs(outs[0][:, 2:] - refq[:, 2:]).max())
            assert np.all(outs[0][np.isnan(xq)] == 0)
    # (1,3,3) / (1,2,2) (maxPool3d_2a / 3a) on 16-bit activations at larger sizes: even and odd map sizes (the TF pad row / column behind
    # the map, the ceil-mode overhang), an all-negative channel whose windows the pad wins, a NaN
    for (n2, c2, d2, h2, w2) in ((2, 32, 2, 21, 40), (1, 16, 1, 16, 16), (1, 24, 3, 33, 9), (1, 8, 2, 40, 70)):
        xb = rs.randn(n2, c2, d2, h2, w2).astype(np.float32)
        xb[0, :, :, :h2 // 2] = np.maximum(xb[0, :, :, :h2 // 2], 0)
        xb[0, 0] = -np.abs(xb[0, 0])                                                # all-negative channel: the pad behind the map wins its windows
        xb[0, 1, 0, 1, 1] = np.nan
        pads2 = [0, 1, 0, 1, 0, 0]
        for dt in (BF16, F16):
            xq = quantize(xb, dt)
            xt = torch.from_numpy(np.nan_to_num(xq, nan=-np.inf)).requires_grad_(True)
            yq = F.max_pool3d(F.pad(xt, pads2), (1, 3, 3), (1, 2, 2), ceil_mode=True)
            gq = quantize(rs.randn(*yq.shape).astype(np.float32), dt)
            yq.backward(torch.from_numpy(gq))
            refq = xt.grad.numpy()
            xe, ge = bk.dev(encode(cl(xq), dt)), bk.dev(encode(np.ascontiguousarray(cl(gq)), dt))
            outs = []
            for direct in (0, 1):
                _capi.set_option(L, "pool_direct", direct)
                try:
                    go = bk.dev(np.full((n2, d2, h2, w2, c2), 3, np.float32))
                    argb = bk.dev(np.zeros(int(np.prod(yq.shape)), np.uint8))
                    assert L.step_maxpool3d_tf_backward_gather(dt, xe.ptr, n2, d2, h2, w2, c2, c2, 0, 1, 3, 3, 1, 2, 2, dt, ge.ptr, 0, go.ptr, argb.ptr, bk.stream) == 0
                    outs.append(uncl(go.get()).copy())
                finally:
                    _capi.set_option(L, "pool_direct", 0)
            assert np.allclose(outs[0], outs[1], rtol=1e-6, atol=1e-6), ((n2, c2, d2, h2, w2), dt, np.abs(outs[0] - outs[1]).max())
            assert np.allclose(outs[0], refq, rtol=1e-6, atol=1e-6), ((n2, c2, d2, h2, w2), dt, np.abs(outs[0] - refq).max())
    # the pool input as a channel slice [8, 16) of a 24-channel buffer (x_cstride / x_coff): same gradient as the dense tensor
    k, s = POOLS[0]
    xw = np.full((N, D, H, W, 24), 9.0, np.float32)
    xw[..., 8:16] = cl(x)
    pads = []
    for kk, ss in zip(reversed(k), reversed(s)):
        a = max(kk - ss, 0)
        pads += [a // 2, a - a // 2]
    xt = torch.from_numpy(x).requires_grad_(True)
    y = F.max_pool3d(F.pad(xt, pads), k, s, ceil_mode=True)
    gy2 = rs.randn(*y.shape).astype(np.float32)
    y.backward(torch.from_numpy(gy2))
    Do, Ho, Wo = y.shape[2:]
    arg2 = bk.dev(np.zeros(N * Do * Ho * Wo * C, np.uint8))
    gs = bk.dev(np.zeros((N, D, H, W, C), np.float32))
    xwd, gyd2 = bk.dev(xw), bk.dev(np.ascontiguousarray(cl(gy2)))          # (named: the buffers must outlive the call)
    assert L.step_maxpool3d_tf_backward_gather(0, xwd.ptr, N, D, H, W, C, 24, 8, k[0], k[1], k[2], s[0], s[1], s[2], 0, gyd2.ptr, 0, gs.ptr,
                                               arg2.ptr, bk.stream) == 0
    assert np.allclose(uncl(gs.get()), xt.grad.numpy(), rtol=1e-6, atol=1e-6)
    assert L.step_maxpool3d_tf_backward_gather(0, None, 0, D, H, W, C, C, 0, 3, 3, 3, 1, 1, 1, 0, None, 0, None, None, bk.stream) == 0              # empty batch
    assert L.step_maxpool3d_tf_backward_gather(1, xe.ptr, N, D, H, W, 12, 12, 0, 3, 3, 3, 1, 1, 1, 1, ge.ptr, 1, go.ptr, arg.ptr, bk.stream) == -4   # C % 8
    assert L.step_maxpool3d_tf_backward_gather(1, xe.ptr, N, D, H, W, C, C, 0, 3, 3, 3, 1, 1, 1, 2, ge.ptr, 1, go.ptr, arg.ptr, bk.stream) < 0     # gy type


def case_pool_golden(bk, golden):
    g = golden("ops_golden")
    x = R.fill_tensor("golden.pool.in", (2, 5, 6, 9, 11), "image").numpy()
    xp = np.zeros((2, 8, 6, 9, 11), np.float32)       # C=5 padded to 8 channels (16-byte lanes)
    xp[:, :5] = x
    for tag, k, s in (("k133s122", (1, 3, 3), (1, 2, 2)), ("k333s222", (3, 3, 3), (2, 2, 2)),
                      ("k333s111", (3, 3, 3), (1, 1, 1)), ("k222s222", (2, 2, 2), (2, 2, 2))):
        ref = g["pool_" + tag]
        xd = bk.dev(cl(xp))
        y = bk.dev(np.zeros((2,) + ref.shape[2:] + (8,), np.float32))
        assert bk.lib.step_maxpool3d_tf(0, xd.ptr, 2, 6, 9, 11, 8, 8, 0, k[0], k[1], k[2], s[0], s[1], s[2], y.ptr, 8, 0, bk.stream) == 0
        assert np.array_equal(uncl(y.get())[:, :5], ref), tag


def case_clip_from_u8(bk, golden):
    """uint8 HWC frames -> normalised [N,T,3,H,W] clip: bit-exact against the numpy arithmetic of the reference's
    ConvertFromInts / SubtractMeans / DivideStds (data/augmentations.py:68-111), all three scale modes."""
    rs = np.random.RandomState(77)
    N, T, H, W = 2, 3, 5, 7
    fr = rs.randint(0, 256, (N, T, H, W, 3)).astype(np.uint8)
    fr[0, 0, 0, 0] = (0, 255, 128)
    mean, std = np.array([0.1, -0.2, 0.3], np.float32), np.array([1.0, 0.5, 2.0], np.float32)
    for scale in (0, 1, 2):
        img = fr.copy()
        if scale == 0:
            ref = img.astype(np.float32)
        elif scale == 1:
            ref = img.astype(np.float32) / 255.
        else:
            ref = np.clip(img, 0, 255).astype(np.float32) * 2 / 255 - 1.
        ref = ref.astype(np.float32)
        ref -= mean
        ref /= std
        ref = np.ascontiguousarray(np.transpose(ref, (0, 1, 4, 2, 3)))
        src = bk.dev(fr)
        out = bk.dev(np.zeros((N, T, 3, H, W), np.float32))
        m = (ctypes.c_float * 3)(*mean.tolist())
        sd = (ctypes.c_float * 3)(*std.tolist())
        assert bk.lib.step_clip_from_u8(src.ptr, N, T, H, W, scale, m, sd, F32, out.ptr, bk.stream) == 0
        assert np.array_equal(out.get(), ref), scale
    outb = bk.dev(np.zeros((N, T, 3, H, W), np.uint16))
    assert bk.lib.step_clip_from_u8(bk.dev(fr).ptr, N, T, H, W, 2, None, None, BF16, outb.ptr, bk.stream) == 0
    ref2 = np.transpose(fr.astype(np.float32) * 2 / 255 - 1., (0, 1, 4, 2, 3))
    assert np.array_equal(outb.get(), to_bf16_bits(ref2))
    assert bk.lib.step_clip_from_u8(None, N, T, H, W, 3, None, None, F32, out.ptr, bk.stream) < 0
    # the 16-pixels-per-thread form (H * W % 16 == 0: every network resolution) against numpy AND against the per-pixel kernel, every
    # dtype and scale mode, a frame count that leaves the last workgroup partial
    N, T, H, W = 2, 5, 12, 20
    fr = rs.randint(0, 256, (N, T, H, W, 3)).astype(np.uint8)
    fr[1, 4, 11, 16:20] = ((0, 255, 128), (255, 0, 1), (254, 127, 129), (1, 2, 3))
    src = bk.dev(fr)
    m = (ctypes.c_float * 3)(*mean.tolist())
    sd = (ctypes.c_float * 3)(*std.tolist())
    for scale in (0, 1, 2):
        ref = fr.astype(np.float32)
        if scale == 1:
            ref = ref / np.float32(255.)
        elif scale == 2:
            ref = ref * np.float32(2) / np.float32(255) - np.float32(1.)
        ref = ((ref.astype(np.float32) - mean) / std).astype(np.float32)
        ref = np.ascontiguousarray(np.transpose(ref, (0, 1, 4, 2, 3)))
        for dt, z in ((F32, np.float32), (BF16, np.uint16), (F16, np.uint16)):
            got = {}
            for vec in (1, 0):
                with _capi.options(bk.lib, clip_vec=vec):
                    o = bk.dev(np.zeros((N, T, 3, H, W), z))
                    assert bk.lib.step_clip_from_u8(src.ptr, N, T, H, W, scale, m, sd, dt, o.ptr, bk.stream) == 0
                    got[vec] = o.get()
            assert np.array_equal(got[1], got[0]), (scale, dt)
            want = ref if dt == F32 else (to_bf16_bits(ref) if dt == BF16 else ref.astype(np.float16).view(np.uint16))
            assert np.array_equal(got[1], want), (scale, dt)


def case_adam_flat(bk, golden):
    """One launch over a flat arena against torch.optim.Adam (what train.py:126 constructs) stepping the same tensors as
    single-tensor groups with their own lr / weight_decay (utils/solver.py:12-93); three steps, gradient scale, fused
    gradient clear.  Tolerance 2e-6 relative to max|p| per step (fp32 rounding order: lerp / addcdiv are fused there)."""
    rs = np.random.RandomState(11)
    sizes = [12, 64, 4, 100, 28, 1000]                      # multiples of 4 (the arena pads every tensor to that)
    lrs = [1e-3, 2e-3, 5e-4, 1e-3, 1e-2, 3e-4]
    wds = [0.0, 1e-2, 0.0, 1e-4, 0.0, 1e-7]
    n = sum(sizes)
    p0 = rs.randn(n).astype(np.float32)
    tp = []
    off = 0
    for sz_ in sizes:
        tp.append(torch.nn.Parameter(torch.from_numpy(p0[off:off + sz_].copy())))
        off += sz_
    opt = torch.optim.Adam([{"params": [t], "lr": lr, "weight_decay": wd} for t, lr, wd in zip(tp, lrs, wds)], lr=1e-3)
    P, M, V = bk.dev(p0.copy()), bk.dev(np.zeros(n, np.float32)), bk.dev(np.zeros(n, np.float32))
    ends = bk.dev(np.cumsum(sizes).astype(np.int64))
    LR, WD = bk.dev(np.array(lrs, np.float32)), bk.dev(np.array(wds, np.float32))
    scale = 0.25
    # the same trajectory with the step counter on the device (step_adam_flat_dev: what a captured training step replays)
    P2, M2, V2 = bk.dev(p0.copy()), bk.dev(np.zeros(n, np.float32)), bk.dev(np.zeros(n, np.float32))    # (EmuBackend buffers alias their array)
    cnt, bc = bk.dev(np.zeros(1, np.int64)), bk.dev(np.zeros(2, np.float32))
    for step_no in (1, 2, 3):
        g = (rs.randn(n) * (10.0 ** rs.uniform(-4, 1, n))).astype(np.float32)
        off = 0
        for t, sz_ in zip(tp, sizes):
            t.grad = torch.from_numpy(g[off:off + sz_] * np.float32(scale))
            off += sz_
        opt.step()
        G, G2 = bk.dev(g.copy()), bk.dev(g.copy())
        zero = int(step_no != 2)
        assert bk.lib.step_adam_flat(P.ptr, G.ptr, M.ptr, V.ptr, n, ends.ptr, LR.ptr, WD.ptr, len(sizes), 0.9, 0.999, 1e-8,
                                     step_no, scale, zero, bk.stream) == 0
        ref = np.concatenate([t.detach().numpy() for t in tp])
        refm = np.concatenate([opt.state[t]["exp_avg"].numpy() for t in tp])
        refv = np.concatenate([opt.state[t]["exp_avg_sq"].numpy() for t in tp])
        assert np.abs(P.get() - ref).max() <= 2e-6 * np.abs(ref).max(), step_no
        ga = np.abs(g * np.float32(scale)) + 1e-2                 # moments: a few ulp of the operands (m may cancel against g)
        assert np.all(np.abs(M.get() - refm) <= 1e-5 * np.abs(refm) + 1e-6 * ga), step_no
        assert np.all(np.abs(V.get() - refv) <= 1e-5 * np.abs(refv) + 1e-6 * ga * ga), step_no
        assert bk.lib.step_adam_flat_dev(P2.ptr, G2.ptr, M2.ptr, V2.ptr, n, ends.ptr, LR.ptr, WD.ptr, len(sizes), 0.9, 0.999, 1e-8,
                                         cnt.ptr, bc.ptr, scale, zero, bk.stream) == 0
        assert int(cnt.get()[0]) == step_no
        assert np.abs(P2.get() - P.get()).max() <= 2e-7 * np.abs(ref).max() and np.array_equal(M2.get(), M.get()) \
            and np.array_equal(V2.get(), V.get()) and np.array_equal(G2.get(), G.get()), step_no
        assert np.array_equal(G.get(), np.zeros(n, np.float32) if zero else g)
    assert bk.lib.step_adam_flat_dev(P2.ptr, G2.ptr, M2.ptr, V2.ptr, n, ends.ptr, LR.ptr, WD.ptr, len(sizes), 0.9, 0.999, 1e-8,
                                     None, bc.ptr, scale, 0, bk.stream) < 0
    assert bk.lib.step_adam_flat(P.ptr, G.ptr, M.ptr, V.ptr, n + 2, ends.ptr, LR.ptr, WD.ptr, len(sizes), 0.9, 0.999, 1e-8, 1, 1.0,
                                 0, bk.stream) < 0                                        # n % 4
    assert bk.lib.step_adam_flat(P.ptr, G.ptr, M.ptr, V.ptr, n, ends.ptr, LR.ptr, WD.ptr, len(sizes), 0.9, 0.999, 1e-8, 0, 1.0,
                                 0, bk.stream) < 0                                        # steps count from 1
    assert bk.lib.step_adam_flat(P.ptr, G.ptr, M.ptr, V.ptr, 0, ends.ptr, LR.ptr, WD.ptr, len(sizes), 0.9, 0.999, 1e-8, 1, 1.0,
                                 0, bk.stream) == 0                                       # empty arena: no launch


def case_adam_flat_amp(bk, golden):
    """step_adam_flat_amp -- dynamic loss scaling on the device (train.py:136-139,342-345: apex amp O1) -- against torch's own pair
    torch.optim.Adam + torch.amp.GradScaler semantics restated on the host: clean steps equal step_adam_flat_dev with grad_scale / scale bit for
    bit and count towards the growth interval; a step whose gradients hold an inf or a nan changes NOTHING but the scale (halved), the
    tracker (reset) and -- when asked -- the cleared gradients; the step count does not advance on a skipped step."""
    rs = np.random.RandomState(5)
    sizes = [64, 8, 256]
    n = sum(sizes)
    p0 = rs.randn(n).astype(np.float32)
    ends = bk.dev(np.cumsum(sizes).astype(np.int64))
    LR, WD = bk.dev(np.array([1e-3, 2e-3, 5e-4], np.float32)), bk.dev(np.array([0.0, 1e-2, 0.0], np.float32))
    P, M, V = bk.dev(p0.copy()), bk.dev(np.zeros(n, np.float32)), bk.dev(np.zeros(n, np.float32))
    P2, M2, V2 = bk.dev(p0.copy()), bk.dev(np.zeros(n, np.float32)), bk.dev(np.zeros(n, np.float32))
    cnt, bc = bk.dev(np.zeros(1, np.int64)), bk.dev(np.zeros(2, np.float32))
    cnt2, bc2 = bk.dev(np.zeros(1, np.int64)), bk.dev(np.zeros(2, np.float32))
    amp = bk.dev(np.array([1024.0, 0.0, 0.0, 0.0], np.float32))
    scale, tracker, steps = 1024.0, 0, 0
    interval = 3
    plan = ["ok", "ok", "inf", "ok", "ok", "ok", "nan", "ok"]          # growth after 3 clean steps in a row; two overflows
    for k, kind in enumerate(plan):
        g = (rs.randn(n) * 0.1).astype(np.float32)
        gs = (g * np.float32(scale)).astype(np.float32)                 # what backward of the scaled loss leaves in the arena
        if kind == "inf":
            gs[17] = np.inf
        elif kind == "nan":
            gs[n - 3] = np.nan
        G = bk.dev(gs.copy())
        before = (P.get().copy(), M.get().copy(), V.get().copy())
        zero = int(k % 2 == 0)
        assert bk.lib.step_adam_flat_amp(P.ptr, G.ptr, M.ptr, V.ptr, n, ends.ptr, LR.ptr, WD.ptr, len(sizes), 0.9, 0.999, 1e-8, cnt.ptr, bc.ptr,
                                         0.5, zero, amp.ptr, 2.0, 0.5, interval, bk.stream) == 0
        if kind == "ok":
            G2 = bk.dev(gs.copy())
            assert bk.lib.step_adam_flat_dev(P2.ptr, G2.ptr, M2.ptr, V2.ptr, n, ends.ptr, LR.ptr, WD.ptr, len(sizes), 0.9, 0.999, 1e-8, cnt2.ptr,
                                             bc2.ptr, 0.5 * (1.0 / scale), zero, bk.stream) == 0
            assert np.array_equal(P.get(), P2.get()) and np.array_equal(M.get(), M2.get()) and np.array_equal(V.get(), V2.get()), k
            steps += 1
            tracker += 1
            if tracker == interval:
                scale, tracker = scale * 2.0, 0
        else:
            assert all(np.array_equal(a, b) for a, b in zip(before, (P.get(), M.get(), V.get()))), k      # skipped: nothing moved
            scale, tracker = scale * 0.5, 0
        assert int(cnt.get()[0]) == steps, (k, cnt.get(), steps)
        st = amp.get()
        assert st[0] == np.float32(scale) and st[1] == np.float32(tracker) and st[2] == 0.0, (k, st, scale, tracker)
        if zero:
            assert not G.get().any(), k
        else:
            assert np.array_equal(G.get(), gs, equal_nan=True), k
    assert bk.lib.step_adam_flat_amp(P.ptr, None, M.ptr, V.ptr, n, ends.ptr, LR.ptr, WD.ptr, len(sizes), 0.9, 0.999, 1e-8, cnt.ptr, bc.ptr,
                                     1.0, 0, amp.ptr, 2.0, 0.5, interval, bk.stream) < 0
    assert bk.lib.step_adam_flat_amp(P.ptr, P.ptr, M.ptr, V.ptr, n, ends.ptr, LR.ptr, WD.ptr, len(sizes), 0.9, 0.999, 1e-8, cnt.ptr, bc.ptr,
                                     1.0, 0, None, 2.0, 0.5, interval, bk.stream) < 0


def case_head_outputs(bk, golden):
    """step_head_outputs / _backward (everything behind TwoBranchNet's last two GEMMs, models/two_branch.py:246-342, as one launch each)
    against the element-wise torch formulation under autograd: frame-mean logits, sigmoid, local / first / last boxes, BCE-with-logits
    x class mask, masked-mean smooth-L1 of the centre / first / last predictions against encode_coef targets; one chunk (Tl = 3: centre,
    first and last frame coincide) and three (Tl = 9); partly and entirely zero masks (zero losses AND zero gradients, the reference's
    `if mask.sum():`); large regression errors (the linear branch of smooth-L1); inference mode (no targets); 16-bit inputs."""
    from step_amd.tube_math import encode_coef
    rs = np.random.RandomState(17)
    NC, T = 60, 3
    for dt, N, Tl, zero_cls, zero_box in ((F32, 7, 3, False, False), (F32, 5, 9, False, False), (F32, 4, 9, True, False), (F32, 4, 3, False, True),
                                          (BF16, 6, 9, False, False), (F16, 3, 3, False, False), (F32, 300, 3, False, False)):
        chunks = Tl // T
        cidx = [j * T + T // 2 for j in range(chunks)]
        logits = quantize((rs.randn(N * Tl, NC) * 2).astype(np.float32), dt)
        reg = quantize((rs.randn(N * Tl, 12) * rs.choice([0.2, 3.0], (N * Tl, 1))).astype(np.float32), dt)
        xy = rs.uniform(0, 250, (N, Tl, 2))
        tubes = np.concatenate([np.zeros((N, Tl, 1)), xy, xy + rs.uniform(20, 140, (N, Tl, 2))], 2).astype(np.float32)
        targets = np.zeros((N, 3, 6 + NC), np.float32)
        gxy = rs.uniform(0, 250, (N, 3, 2))
        targets[:, :, :4] = np.concatenate([gxy, gxy + rs.uniform(20, 140, (N, 3, 2))], 2)
        targets[:, :, 4] = (rs.rand(N, 3) < 0.7)
        targets[:, :, 5] = (rs.rand(N, 3) < 0.7)
        targets[:, :, 6:] = (rs.rand(N, 3, NC) < 0.1)
        if zero_cls:
            targets[:, :, 4] = 0
        if zero_box:
            targets[:, :, 5] = 0
        # ---- the reference formulation (as heads.py's element-wise path, SYNC_FREE_LOSSES)
        lt, rt = torch.from_numpy(logits).requires_grad_(True), torch.from_numpy(reg).requires_grad_(True)
        tb, tg = torch.from_numpy(tubes), torch.from_numpy(targets)
        gc = lt.reshape(N, Tl, NC).mean(1)
        r3 = rt.reshape(N, Tl, 12)
        ll = r3[..., 0:4]
        lo, lo2 = cidx[0] - T // 2, cidx[-1] - T // 2
        fl = ll[:, lo:lo + T] + r3[:, lo:lo + T, 4:8]
        la = ll[:, lo2:lo2 + T] + r3[:, lo2:lo2 + T, 8:12]
        ct, ft, ltg = tg[:, 1], tg[:, 0], tg[:, -1]
        m = ct[:, 4].reshape(-1, 1)
        pos = (m.sum() > 0).float()
        l_cls = F.binary_cross_entropy_with_logits(gc, ct[:, 6:] * m, reduction="none") * pos

        def mean_over(l, mk):
            s_ = mk.sum()
            return (l * mk).sum() / (s_ if float(s_) > 0 else 1.0)
        l_loc = mean_over(F.smooth_l1_loss(ll[:, cidx[chunks // 2]], encode_coef(ct[:, :4], tb[:, cidx[chunks // 2], 1:]), reduction="none"),
                          ct[:, 5].reshape(-1, 1).repeat(1, 4))
        ntgt = encode_coef(torch.cat([ft[:, :4], ltg[:, :4]], 0), torch.cat([tb[:, cidx[0], 1:], tb[:, cidx[-1], 1:]], 0))
        l_nbr = mean_over(F.smooth_l1_loss(torch.cat([fl[:, T // 2], la[:, T // 2]], 0), ntgt, reduction="none"),
                          torch.cat([ft[:, 5].reshape(-1, 1).repeat(1, 4), ltg[:, 5].reshape(-1, 1).repeat(1, 4)], 0))
        wc = torch.from_numpy(rs.randn(N, NC).astype(np.float32))
        (l_cls * wc).sum().add(1.7 * l_loc).add(-0.6 * l_nbr).backward()
        # ---- the kernels
        L_, R_ = bk.dev(encode(logits, dt)), bk.dev(encode(reg, dt))
        TB, TG = bk.dev(tubes), bk.dev(targets)
        prob, oll, ofl, ola = bk.dev(np.zeros((N, NC), np.float32)), bk.dev(np.zeros((N, Tl, 4), np.float32)), bk.dev(np.zeros((N, T, 4), np.float32)), bk.dev(np.zeros((N, T, 4), np.float32))
        lc, lo_, ln = bk.dev(np.zeros((N, NC), np.float32)), bk.dev(np.zeros(1, np.float32)), bk.dev(np.zeros(1, np.float32))
        assert bk.lib.step_head_outputs(dt, L_.ptr, NC, R_.ptr, 12, N, Tl, T, NC, TB.ptr, TG.ptr, prob.ptr, oll.ptr, ofl.ptr, ola.ptr, lc.ptr, lo_.ptr, ln.ptr,
                                        bk.stream) == 0
        close = lambda a, b, t_=2e-6: np.abs(a - b).max() <= t_ * max(1.0, np.abs(b).max())
        assert close(prob.get(), torch.sigmoid(gc).detach().numpy()) and close(oll.get(), ll.detach().numpy(), 0) and close(ofl.get(), fl.detach().numpy())
        assert close(ola.get(), la.detach().numpy()) and close(lc.get(), l_cls.detach().numpy(), 5e-6)
        assert abs(float(lo_.get()[0]) - float(l_loc.detach())) <= 5e-6 * max(1.0, abs(float(l_loc.detach()))) and abs(float(ln.get()[0]) - float(l_nbr.detach())) <= 5e-6 * max(1.0, abs(float(l_nbr.detach())))
        if zero_cls:
            assert not lc.get().any()
        if zero_box:
            assert float(lo_.get()[0]) == 0.0 and float(ln.get()[0]) == 0.0
        GL, GR = bk.dev(np.full((N * Tl, NC), 7, NP_DT[dt])), bk.dev(np.full((N * Tl, 12), 7, NP_DT[dt]))
        gcl, glo, gnb = bk.dev(wc.numpy()), bk.dev(np.array([1.7], np.float32)), bk.dev(np.array([-0.6], np.float32))
        outs = []
        for _ in range(2):
            assert bk.lib.step_head_outputs_backward(dt, L_.ptr, NC, R_.ptr, 12, N, Tl, T, NC, TB.ptr, TG.ptr, gcl.ptr, glo.ptr, gnb.ptr, GL.ptr, GR.ptr,
                                                     bk.stream) == 0
            outs.append((GL.get().copy(), GR.get().copy()))
        assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
        gl_, gr_ = decode(outs[0][0], dt), decode(outs[0][1], dt)
        t_ = 2e-6 if dt == F32 else tol(dt)
        assert np.abs(gl_ - lt.grad.numpy()).max() <= t_ * max(1e-6, np.abs(lt.grad.numpy()).max()) + 1e-9, (dt, N, Tl)
        assert np.abs(gr_ - rt.grad.numpy()).max() <= t_ * max(1e-6, np.abs(rt.grad.numpy()).max()) + 1e-9, (dt, N, Tl)
        if zero_cls:
            assert not gl_.any()
        if zero_box:
            assert not gr_.any()
    # inference: no targets, no tubes -- probabilities and boxes, three zero losses
    z3 = bk.dev(np.full(3, 5.0, np.float32))
    assert bk.lib.step_head_outputs(dt, L_.ptr, NC, R_.ptr, 12, N, Tl, T, NC, None, None, prob.ptr, oll.ptr, ofl.ptr, ola.ptr, z3.ptr, z3.ptr + 4 if isinstance(z3.ptr, int) else ctypes.c_void_p(z3.ptr.value + 4),
                                    z3.ptr + 8 if isinstance(z3.ptr, int) else ctypes.c_void_p(z3.ptr.value + 8), bk.stream) == 0
    assert not z3.get().any() and close(prob.get(), torch.sigmoid(gc).detach().numpy())
    assert bk.lib.step_head_outputs(dt, L_.ptr, NC, R_.ptr, 12, N, 4, T, NC, None, None, prob.ptr, oll.ptr, ofl.ptr, ola.ptr, z3.ptr, z3.ptr, z3.ptr, bk.stream) == -2   # Tl % T


def case_bn_train(bk, golden):
    """step_bn_train_forward / _backward (batch-statistics BatchNorm + ReLU of --freeze_stats False, models/i3dpt.py:95-110) against
    torch's own F.batch_norm(training=True) + relu under autograd on the same (storage-rounded) input: output, saved statistics, the
    running-statistics update (momentum 0.1, unbiased variance), gz / ggamma / gbeta; channel-slice strides, several chunks with a
    ragged tail, a large |mean| / std ratio (shifted sums), ReLU on and off, fp32 and 16-bit gradients; bit-reproducible."""
    rs = np.random.RandomState(9)
    for dt, M, C, zcs, relu, gdt in ((F32, 5000, 24, 24, 1, F32), (F32, 300, 8, 16, 0, F32), (BF16, 4500, 16, 24, 1, BF16), (F16, 2100, 8, 8, 1, F32)):
        z = (rs.randn(M, C) * rs.uniform(0.5, 2.0, C) + rs.uniform(-30, 30, C)).astype(np.float32)
        gamma, beta = rs.uniform(0.5, 1.5, C).astype(np.float32), rs.uniform(-1, 1, C).astype(np.float32)
        rm0, rv0 = rs.randn(C).astype(np.float32), rs.uniform(0.5, 2, C).astype(np.float32)
        gy = rs.randn(M, C).astype(np.float32)
        zq, gq = quantize(z, dt), quantize(gy, gdt)
        zt = torch.from_numpy(zq).requires_grad_(True)
        gt, bt = torch.from_numpy(gamma).requires_grad_(True), torch.from_numpy(beta).requires_grad_(True)
        rm, rv = torch.from_numpy(rm0.copy()), torch.from_numpy(rv0.copy())
        yr = F.batch_norm(zt.t().reshape(1, C, M), rm, rv, gt, bt, True, 0.1, 1e-5).reshape(C, M).t()
        yr = F.relu(yr) if relu else yr
        (yr * torch.from_numpy(gq)).sum().backward()        # (the kernel masks with the STORED output; rounding never moves a positive y to <= 0)
        zbuf = np.zeros((M, zcs), np.float32)
        zbuf[:, :C] = zq
        Z = bk.dev(encode(zbuf, dt))
        Y = bk.dev(np.zeros((M, zcs), NP_DT[dt]))
        G, Bt = bk.dev(gamma), bk.dev(beta)
        RM, RV = bk.dev(rm0.copy()), bk.dev(rv0.copy())
        SM, SI = bk.dev(np.zeros(C, np.float32)), bk.dev(np.zeros(C, np.float32))
        wsb = bk.lib.step_bn_train_workspace_bytes(M, C)
        WS = bk.dev(np.zeros(wsb // 4 + 4, np.float32))
        assert bk.lib.step_bn_train_forward(dt, Z.ptr, zcs, M, C, G.ptr, Bt.ptr, 1e-5, 0.1, RM.ptr, RV.ptr, SM.ptr, SI.ptr, relu, Y.ptr, zcs, WS.ptr, wsb,
                                            bk.stream) == 0
        y = decode(Y.get(), dt)[:, :C]
        t_ = tol(dt)
        assert np.abs(y - yr.detach().numpy()).max() <= t_ * max(1.0, np.abs(yr.detach().numpy()).max()) + 2e-5, (dt, M, C)
        mean = zq.astype(np.float64).mean(0)
        var = zq.astype(np.float64).var(0)
        assert np.abs(SM.get() - mean).max() <= 1e-6 * np.abs(mean).max() + 1e-6
        assert np.abs(SI.get() - 1 / np.sqrt(var + 1e-5)).max() <= 2e-6 * (1 / np.sqrt(var + 1e-5)).max()
        assert np.abs(RM.get() - rm.numpy()).max() <= 1e-6 * max(1.0, np.abs(rm.numpy()).max())
        assert np.abs(RV.get() - rv.numpy()).max() <= 2e-6 * max(1.0, np.abs(rv.numpy()).max())
        GY = bk.dev(encode(gq, gdt))
        GZ = bk.dev(np.zeros((M, C), NP_DT[dt]))
        GG, GB = bk.dev(np.zeros(C, np.float32)), bk.dev(np.zeros(C, np.float32))
        outs = []
        for _ in range(2):
            assert bk.lib.step_bn_train_backward(dt, Z.ptr, zcs, Y.ptr, zcs, gdt, GY.ptr, C, M, C, relu, G.ptr, SM.ptr, SI.ptr, GZ.ptr, GG.ptr, GB.ptr,
                                                 WS.ptr, wsb, bk.stream) == 0
            outs.append((GZ.get().copy(), GG.get().copy(), GB.get().copy()))
        assert all(np.array_equal(a, b) for a, b in zip(*outs))
        gz = decode(outs[0][0], dt)
        ref_gz = zt.grad.numpy()
        assert np.abs(gz - ref_gz).max() <= max(t_, 2e-5) * max(1e-3, np.abs(ref_gz).max()) * (4 if dt != F32 else 1), (dt, M, C, np.abs(gz - ref_gz).max(), np.abs(ref_gz).max())
        assert np.abs(outs[0][1] - gt.grad.numpy()).max() <= (2e-2 if dt != F32 else 2e-4) * max(1.0, np.abs(gt.grad.numpy()).max()), (dt, M)
        assert np.abs(outs[0][2] - bt.grad.numpy()).max() <= (2e-2 if dt != F32 else 2e-4) * max(1.0, np.abs(bt.grad.numpy()).max()), (dt, M)
    assert bk.lib.step_bn_train_forward(F32, Z.ptr, 6, 10, 6, None, None, 1e-5, 0.1, None, None, SM.ptr, SI.ptr, 1, Y.ptr, 6, WS.ptr, wsb, bk.stream) == -4   # C % 4
    assert bk.lib.step_bn_train_forward(F32, Z.ptr, 8, 0, 8, None, None, 1e-5, 0.1, None, None, SM.ptr, SI.ptr, 1, Y.ptr, 8, WS.ptr, wsb, bk.stream) == 0     # empty batch


def case_pack_weight_dgrad(bk, golden):
    """step_conv_pack_weight_dgrad == step_conv_pack_weight of the flipped / transposed / zero-padded weight, bit for bit."""
    rs = np.random.RandomState(17)
    for (Cout, Cin, k, pad) in ((40, 24, (3, 3, 3), 0), (12, 72, (1, 1, 1), 4), (20, 33, (1, 3, 3), 12)):
        w = rs.randn(Cout, Cin, *k).astype(np.float32)
        wt = np.ascontiguousarray(np.flip(w, (2, 3, 4)).transpose(1, 0, 2, 3, 4))           # [Cin, Cout, ...]
        wt = np.concatenate([wt, np.zeros((Cin, pad) + k, np.float32)], 1) if pad else wt
        for dt in (F32, BF16, F16):
            n = bk.lib.step_conv_packed_elems(Cin, Cout + pad, *k)
            a = bk.dev(np.zeros(n, np.float32 if dt == F32 else np.uint16))
            b = bk.dev(np.zeros(n, np.float32 if dt == F32 else np.uint16))
            assert bk.lib.step_conv_pack_weight(bk.dev(np.ascontiguousarray(wt)).ptr, Cin, Cout + pad, k[0], k[1], k[2], dt, None, a.ptr, bk.stream) == 0
            assert bk.lib.step_conv_pack_weight_dgrad(bk.dev(w).ptr, Cout, Cin, k[0], k[1], k[2], dt, Cout + pad, b.ptr, bk.stream) == 0
            assert np.array_equal(a.get(), b.get()), (Cout, Cin, k, dt)
    w = bk.dev(np.zeros((4, 4, 1, 2, 2), np.float32))
    assert bk.lib.step_conv_pack_weight_dgrad(w.ptr, 4, 4, 1, 2, 2, F32, 4, w.ptr, bk.stream) < 0       # even kernel
    assert bk.lib.step_conv_pack_weight_dgrad(w.ptr, 4, 4, 1, 1, 1, F32, 3, w.ptr, bk.stream) < 0       # cin_pad < Cout


def case_pack_weights_one_launch(bk, golden):
    """step_conv_pack_weights (every weight of a net in one launch) == the single-weight entries on the effective weight
    w[:, lo + perm[j]], forward and data-gradient images, bit for bit."""
    import ctypes
    from step_amd import _capi
    rs = np.random.RandomState(23)
    # (Cout, w_cin, k, (lo, hi) | None, permuted?, dgrad pad | None)
    specs = ((40, 24, (3, 3, 3), None, False, None), (40, 24, (3, 3, 3), None, False, 0), (12, 72, (1, 1, 1), (8, 56), False, None),
             (12, 72, (1, 1, 1), (8, 56), True, 4), (20, 33, (1, 3, 3), None, True, None), (20, 33, (1, 3, 3), (1, 33), False, 12),
             (64, 16, (1, 1, 1), None, False, None), (7, 5, (3, 1, 1), None, False, 1))
    for dt in (F32, BF16, F16):
        items = (_capi.PackItem * len(specs))()
        keep, want, outs = [], [], []
        for j, (Cout, wcin, k, sl, permuted, dpad) in enumerate(specs):
            w = rs.randn(Cout, wcin, *k).astype(np.float32)
            lo, hi = sl if sl else (0, wcin)
            cin = hi - lo
            perm = rs.permutation(cin).astype(np.int32) if permuted else None
            weff = w[:, lo:hi]
            weff = np.ascontiguousarray(weff[:, perm] if perm is not None else weff)
            wd, pd = bk.dev(w), bk.dev(perm)
            if dpad is None:
                n = bk.lib.step_conv_packed_elems(Cout, cin, *k)
                a = bk.dev(np.zeros(n, NP_DT[dt]))
                assert bk.lib.step_conv_pack_weight(bk.dev(weff).ptr, Cout, cin, k[0], k[1], k[2], dt, None, a.ptr, bk.stream) == 0
            else:
                n = bk.lib.step_conv_packed_elems(cin, Cout + dpad, *k)
                a = bk.dev(np.zeros(n, NP_DT[dt]))
                assert bk.lib.step_conv_pack_weight_dgrad(bk.dev(weff).ptr, Cout, cin, k[0], k[1], k[2], dt, Cout + dpad, a.ptr, bk.stream) == 0
            b = bk.dev(np.full(n, 7, NP_DT[dt]))
            it = items[j]
            it.w, it.perm_c, it.packed = wd.ptr, pd.ptr, b.ptr
            it.Cout, it.Cin, it.w_cin, it.cin_lo = Cout, cin, wcin, lo
            it.kd, it.kh, it.kw = k
            it.dgrad, it.cin_pad = (0, 0) if dpad is None else (1, Cout + dpad)
            keep += [wd, pd]
            want.append(a)
            outs.append(b)
        table = bk.dev(np.frombuffer(bytes(items), np.uint8).copy())
        assert bk.lib.step_conv_pack_weights(table.ptr, len(specs), dt, bk.stream) == 0
        for j, (a, b) in enumerate(zip(want, outs)):
            assert np.array_equal(a.get(), b.get()), (dt, specs[j])
    assert bk.lib.step_conv_pack_weights(None, 0, F32, bk.stream) == 0
    assert bk.lib.step_conv_pack_weights(None, 2, F32, bk.stream) < 0
    assert bk.lib.step_conv_pack_weights(table.ptr, 1, 9, bk.stream) < 0


def case_act_grad(bk, golden):
    """g = gy * (y > 0) * scale[c] against the torch element-wise chain of the unit's backward (cast, mask, multiply, cast):
    bit-exact in fp32 and after the rounding to the 16-bit activation type; both outputs, either alone, no relu / no scale."""
    rs = np.random.RandomState(21)
    M, C = 37, 24
    y = rs.randn(M, C).astype(np.float32)
    y[::5] = 0.0                                                 # y == 0 is masked (strict >)
    gy = rs.randn(M, C).astype(np.float32)
    scale = (1 + 0.3 * rs.randn(C)).astype(np.float32)
    for dt in (F32, BF16, F16):
        yq, gq = quantize(y, dt), quantize(gy, dt)
        for g_dt, gsrc in ((F32, gy), (dt, gq)):
            for relu in (1, 0):
                for sc in (scale, None):
                    ref = gsrc.copy()
                    if sc is not None:
                        ref = ref * sc[None, :]
                    if relu:
                        ref = ref * (yq > 0)
                    ref = ref.astype(np.float32)
                    yd, gd = bk.dev(encode(yq, dt)), bk.dev(encode(gsrc, g_dt))
                    o32 = bk.dev(np.full((M, C), 9.0, np.float32))
                    oT = bk.dev(encode(np.full((M, C), 9.0, np.float32), dt))
                    sd = bk.dev(sc) if sc is not None else None
                    assert bk.lib.step_act_grad(dt, yd.ptr, 0, g_dt, gd.ptr, 0, sd.ptr if sd else None, M, C, relu, o32.ptr, oT.ptr, bk.stream) == 0
                    assert np.array_equal(o32.get(), ref), (dt, g_dt, relu)
                    assert np.array_equal(decode(oT.get(), dt), quantize(ref, dt)), (dt, g_dt, relu)
    # 16-bit, 16-byte vectors per lane with the (pixel, channel) pair carried along the grid-stride walk: enough vectors for several
    # trips (the emulator build caps the grid at two workgroups; on the GPU the walk starts at 2 M vectors), y and gy as channel
    # slices of wider buffers; C = 12 keeps the 4-wide form
    for (M2, C2, wide) in ((700, 24, 40), (2100, 8, 24), (300, 12, 12)) + (((1 << 18) + 3, 72, 80),) * (bk.name == "gfx950"):
        yw, gw = rs.randn(M2, wide).astype(np.float32), rs.randn(M2, wide).astype(np.float32)
        yw[::3] = 0.0
        sc2 = (1 + 0.3 * rs.randn(C2)).astype(np.float32)
        for dt in (BF16, F16):
            yq, gq = quantize(yw, dt), quantize(gw, dt)
            esz = 2
            yd2, gd2, sd2 = bk.dev(encode(yq, dt)), bk.dev(encode(gq, dt)), bk.dev(sc2)
            yo, go = wide - C2 - (wide - C2) % 8, 0 if wide == C2 else 8
            ref = (gq[:, go:go + C2] * sc2[None, :]).astype(np.float32) * (yq[:, yo:yo + C2] > 0)
            o32b = bk.dev(np.full((M2, C2), 9.0, np.float32))
            oTb = bk.dev(encode(np.full((M2, C2), 9.0, np.float32), dt))
            offp = lambda p_, nbytes: ctypes.c_void_p((p_.value if isinstance(p_, ctypes.c_void_p) else int(p_)) + nbytes)
            assert bk.lib.step_act_grad(dt, offp(yd2.ptr, yo * esz), wide, dt, offp(gd2.ptr, go * esz), wide, sd2.ptr, M2, C2, 1, o32b.ptr, oTb.ptr, bk.stream) == 0
            assert np.array_equal(o32b.get(), ref.astype(np.float32)), (M2, C2, dt)
            assert np.array_equal(decode(oTb.get(), dt), quantize(ref.astype(np.float32), dt)), (M2, C2, dt)
    yd, gd = bk.dev(y), bk.dev(gy)
    o32 = bk.dev(np.zeros((M, C), np.float32))
    assert bk.lib.step_act_grad(F32, yd.ptr, 0, F32, gd.ptr, 0, None, M, C, 1, o32.ptr, None, bk.stream) == 0      # fp32 output only
    assert np.array_equal(o32.get(), gy * (y > 0))
    # channel slices: y = columns [8, 16) of a 24-wide buffer, gy = columns [4, 12) of the other one
    o8 = bk.dev(np.zeros((M, 8), np.float32))
    s8 = bk.dev(scale[:8].copy())
    off = lambda p_, nbytes: ctypes.c_void_p((p_.value if isinstance(p_, ctypes.c_void_p) else int(p_)) + nbytes)
    assert bk.lib.step_act_grad(F32, off(yd.ptr, 8 * 4), C, F32, off(gd.ptr, 4 * 4), C, s8.ptr, M, 8, 1, o8.ptr, None, bk.stream) == 0
    assert np.array_equal(o8.get(), (gy[:, 4:12] * scale[None, :8] * (y[:, 8:16] > 0)).astype(np.float32))
    assert bk.lib.step_act_grad(F32, yd.ptr, 0, F32, gd.ptr, 0, None, M, 22, 1, o32.ptr, None, bk.stream) == -4     # C % 4: caller falls back
    assert bk.lib.step_act_grad(F32, yd.ptr, 0, BF16, gd.ptr, 0, None, M, C, 1, o32.ptr, None, bk.stream) < 0
    assert bk.lib.step_act_grad(F32, None, 0, F32, gd.ptr, 0, None, M, C, 1, o32.ptr, None, bk.stream) < 0
    assert bk.lib.step_act_grad(F32, yd.ptr, 8, F32, gd.ptr, 0, None, M, C, 1, o32.ptr, None, bk.stream) < 0        # stride < C
    assert bk.lib.step_act_grad(F32, yd.ptr, 0, F32, gd.ptr, 0, None, 0, C, 1, o32.ptr, None, bk.stream) == 0


def case_avgpool_hw(bk, golden):
    rs = np.random.RandomState(6)
    x = rs.randn(2, 8, 3, 13, 13).astype(np.float32)
    ref = F.avg_pool3d(torch.from_numpy(x), (1, 13, 13), (1, 1, 1)).numpy()
    xcl, y = bk.dev(cl(x)), bk.dev(np.zeros((2, 3, 1, 1, 8), np.float32))
    assert bk.lib.step_avgpool_hw(0, xcl.ptr, 2, 3, 13, 13, 8, 13, 13, y.ptr, bk.stream) == 0
    assert np.abs(uncl(y.get()) - ref).max() < 1e-6


def case_transpose_cs(bk, golden):
    rs = np.random.RandomState(7)
    x = rs.randn(2, 37, 50).astype(np.float32)
    xd, y = bk.dev(x), bk.dev(np.zeros((2, 50, 37), np.float32))
    assert bk.lib.step_transpose_cs(xd.ptr, 0, y.ptr, 0, 2, 37, 50, 1, bk.stream) == 0
    assert np.array_equal(y.get(), np.transpose(x, (0, 2, 1)))
    z = bk.dev(np.zeros_like(x))
    assert bk.lib.step_transpose_cs(y.ptr, 0, z.ptr, 0, 2, 37, 50, 0, bk.stream) == 0
    assert np.array_equal(z.get(), x)
    yb = bk.dev(np.zeros((2, 50, 37), np.uint16))
    assert bk.lib.step_transpose_cs(xd.ptr, 0, yb.ptr, 1, 2, 37, 50, 1, bk.stream) == 0
    assert np.array_equal(yb.get(), to_bf16_bits(np.transpose(x, (0, 2, 1))))


CONV_CASES = [
    # (N, Cin, Cout, D, H, W, kernel)
    (1, 16, 32, 2, 8, 16, (3, 3, 3)),       # exactly one 8x16 tile, one half-empty slab, one n-block
    (2, 24, 48, 3, 9, 7, (3, 3, 3)),        # odd sizes, Cin/Cout not multiples of 32
    (1, 40, 72, 2, 5, 37, (3, 3, 3)),       # wide tile shape (4x32), 2 slabs, 3 n-blocks
    (1, 96, 208, 1, 14, 14, (3, 3, 3)),     # an I3D 4b shape: 7 n-blocks
    (2, 32, 40, 3, 7, 7, (1, 3, 3)),        # 2-D 3x3 conv (frames on D)
    (2, 72, 100, 2, 5, 9, (1, 1, 1)),       # pointwise: flat tiles, tail tile
    (1, 200, 16, 1, 13, 13, (1, 1, 1)),     # 7 slabs
    (2, 264, 40, 1, 5, 9, (1, 1, 1)),       # deep pointwise: 128-channel slabs, ragged last slab (264 = 2*128 + 8)
    (1, 64, 40, 9, 7, 7, (1, 3, 3)),        # plane-folded tiles (4 planes x 8x8): 2-D conv over 9 frames, ragged last tile (9 = 2*4 + 1)
    (2, 64, 32, 5, 7, 6, (3, 3, 3)),        # plane-folded tiles with a 3-D kernel: 6-plane halo, clip boundary inside the grid
    (1, 64, 40, 3, 9, 28, (3, 3, 3)),       # general box tiles (28 wide: 1x9x28 = 252 pixels, rows past the box masked)
    (1, 72, 32, 9, 7, 7, (1, 3, 3)),        # general box 5x7x7 on 7x7 maps (2-D conv over 9 frames), ragged last box
]


def _conv_case(bk, case, dts):
    N, Cin, Cout, D, H, W, k = case
    rs = np.random.RandomState(Cin * 7 + Cout)
    x = rs.randn(N, Cin, D, H, W).astype(np.float32)
    w = (rs.randn(Cout, Cin, *k) / np.sqrt(Cin * np.prod(k))).astype(np.float32)
    scale = (1 + 0.1 * rs.randn(Cout)).astype(np.float32)
    shift = (0.2 * rs.randn(Cout)).astype(np.float32)
    for dt in dts:
        got = run_conv(bk, x, w, scale, shift, dt, x_pad=(8, 8), y_pad=(16, 8))
        ref = ref_conv(x, w, scale, shift, dt)
        err = np.abs(got - ref).max() / np.abs(ref).max()
        assert err < tol(dt), (case, dt, err)


def case_conv_units(bk, golden):
    for case in CONV_CASES:
        _conv_case(bk, case, (F32, BF16))


def case_conv_units_four_wave_form(bk, golden):
    """The same shapes through the FOUR-wave form of conv_tap_kernel (128-pixel tiles: 8x16, 2 planes x 8x8, general boxes
    <= 128 pixels, one tap per barrier; option conv_waves = 4 makes the planner choose it wherever the tap kernel runs), plus
    shapes that make each of its tile kinds ragged."""
    with _capi.options(bk.lib, conv_waves=4):
        d = _capi.ConvDesc(dtype=BF16, N=1, D=2, H=8, W=16, Cin=64, Cout=96, kd=3, kh=3, kw=3, x_cstride=64, x_coff=0, y_cstride=96,
                           y_coff=0, res_cstride=0, res_coff=0, relu=1, split=0, y2_cstride=0, y2_coff=0)
        buf = ctypes.create_string_buffer(256)
        assert bk.lib.step_conv_kernel_name(ctypes.byref(d), buf, 256) == 0
        assert buf.value.decode().endswith(", 2, 4, 0>(step::ConvParams)"), buf.value      # ..., TPS, MB = 2, WV = 4, PH = 0>
        extra = [(1, 64, 192, 3, 9, 17, (3, 3, 3)),     # NB = 3: 8x16 tiles, ragged in H and W, two slabs
                 (1, 96, 130, 5, 8, 8, (3, 3, 3)),      # 2 planes x 8x8, ragged plane pairs (5 = 2*2 + 1), 5 channel blocks
                 (2, 32, 64, 2, 6, 21, (1, 3, 3))]      # 2-D kernel, general box on a 6x21 map
        for case in CONV_CASES + extra:
            if case[6] != (1, 1, 1):
                _conv_case(bk, case, (F32, BF16))
        # the streaming pointwise GEMM in its four-wave form (128-pixel tiles), every accumulator depth; NB = 3 runs the
        # two-deep register ring
        pw = [(1, 136, 200, 1, 40, 52, (1, 1, 1)),      # ragged K (136 = 4*32 + 8), ragged last pixel tile, 7 channel blocks
              (2, 128, 128, 2, 25, 41, (1, 1, 1))]      # whole slabs, tail tile
        for nb in (1, 2, 3):
          with _capi.options(bk.lib, conv_nb=nb):
            for case in pw:
                N, Cin, Cout, D, H, W, k = case
                d = _capi.ConvDesc(dtype=BF16, N=N, D=D, H=H, W=W, Cin=Cin, Cout=Cout, kd=1, kh=1, kw=1, x_cstride=Cin, x_coff=0,
                                   y_cstride=Cout, y_coff=0, res_cstride=0, res_coff=0, relu=1, split=0, y2_cstride=0, y2_coff=0)
                assert bk.lib.step_conv_kernel_name(ctypes.byref(d), buf, 256) == 0
                assert ("conv_pw_kernel<step::bf16_t, %d, 4>" % nb) in buf.value.decode(), buf.value
                _conv_case(bk, case, (F32, BF16))


def case_conv_units_two_phase_form(bk, golden):
    """The TWO-PHASE form of conv_tap_kernel (anti-phase wave groups: a step is a load phase and a multiply phase separated
    by barriers, the two channel-half groups of the workgroup run them alternately; option conv_phased >= 1 -- 2 is the default -- selects it wherever
    the 8-wave tap kernel runs in 16-bit storage): same shapes as the classic form, checked against the fp32 reference AND
    bit for bit against the classic form (the accumulation order is the same)."""
    buf = ctypes.create_string_buffer(256)
    extra = [(1, 96, 192, 3, 9, 17, (3, 3, 3)),      # NB = 3, three slabs (two halo re-stagings with the groups realigned), ragged tiles
             (1, 40, 130, 5, 8, 8, (3, 3, 3)),       # 4 planes x 8x8, ragged last slab (40 = 32 + 8), 5 channel blocks (odd: a duplicate block)
             (2, 32, 64, 2, 6, 21, (1, 3, 3))]       # 2-D kernel (10 padded taps: an ODD number of steps per slab), general box
    cases = [c for c in CONV_CASES + extra if c[6] != (1, 1, 1)]
    with _capi.options(bk.lib, conv_waves=8):         # (the planner sends grids this small to the four-wave form)
        classic = {}
        with _capi.options(bk.lib, conv_phased=0):
            for case in cases:
                N, Cin, Cout, D, H, W, k = case
                rs = np.random.RandomState(Cin * 7 + Cout)
                x = rs.randn(N, Cin, D, H, W).astype(np.float32)
                w = (rs.randn(Cout, Cin, *k) / np.sqrt(Cin * np.prod(k))).astype(np.float32)
                scale = (1 + 0.1 * rs.randn(Cout)).astype(np.float32)
                shift = (0.2 * rs.randn(Cout)).astype(np.float32)
                classic[case] = (x, w, scale, shift, run_conv(bk, x, w, scale, shift, BF16, x_pad=(8, 8), y_pad=(16, 8)))
        with _capi.options(bk.lib, conv_phased=1):
            seen = 0
            for case in cases:
                N, Cin, Cout, D, H, W, k = case
                x, w, scale, shift, y0 = classic[case]
                d = _capi.ConvDesc(dtype=BF16, N=N, D=D, H=H, W=W, Cin=Cin, Cout=Cout, kd=k[0], kh=k[1], kw=k[2], x_cstride=Cin, x_coff=0,
                                   y_cstride=Cout, y_coff=0, res_cstride=0, res_coff=0, relu=1, split=0, y2_cstride=0, y2_coff=0)
                assert bk.lib.step_conv_kernel_name(ctypes.byref(d), buf, 256) == 0
                name = buf.value.decode()
                if "conv_tap_kernel" in name and k == (3, 3, 3):
                    assert name.endswith(", 2, 2, 8, 1>(step::ConvParams)"), name
                    seen += 1
                got = run_conv(bk, x, w, scale, shift, BF16, x_pad=(8, 8), y_pad=(16, 8))
                ref = ref_conv(x, w, scale, shift, BF16)
                err = np.abs(got - ref).max() / np.abs(ref).max()
                assert err < tol(BF16), (case, err)
                assert np.array_equal(got, y0), (case, float(np.abs(got - y0).max()))
            assert seen >= 4, seen
            _conv_case(bk, extra[0], (F16,))
        # conv_phased = 2 (round 6): the two-phase form for 1x3x3 windows too (general boxes, NB <= 2: five steps per slab, the last a single tap,
        # ring parity alternating per slab) -- bit for bit the classic form
        with _capi.options(bk.lib, conv_phased=2):
            seen2 = 0
            for case in [c for c in cases if c[6] == (1, 3, 3)] + [(3, 72, 100, 1, 7, 7, (1, 3, 3))]:      # + the heads' plane-folded 7x7 maps, three slabs (odd count: both parities)
                N, Cin, Cout, D, H, W, k = case
                if case in classic:
                    x, w, scale, shift, y0 = classic[case]
                else:
                    rs = np.random.RandomState(Cin * 7 + Cout)
                    x = rs.randn(N, Cin, D, H, W).astype(np.float32)
                    w = (rs.randn(Cout, Cin, *k) / np.sqrt(Cin * np.prod(k))).astype(np.float32)
                    scale = (1 + 0.1 * rs.randn(Cout)).astype(np.float32)
                    shift = (0.2 * rs.randn(Cout)).astype(np.float32)
                    with _capi.options(bk.lib, conv_phased=0):
                        y0 = run_conv(bk, x, w, scale, shift, BF16, x_pad=(8, 8), y_pad=(16, 8))
                d = _capi.ConvDesc(dtype=BF16, N=N, D=D, H=H, W=W, Cin=Cin, Cout=Cout, kd=1, kh=3, kw=3, x_cstride=Cin, x_coff=0,
                                   y_cstride=Cout, y_coff=0, res_cstride=0, res_coff=0, relu=1, split=0, y2_cstride=0, y2_coff=0)
                info = (ctypes.c_int * 10)()
                assert bk.lib.step_conv_plan_info(ctypes.byref(d), info, 10) == 0
                if info[0] == 1 and info[1] == 0 and info[2] <= 2 and info[3] == 8:
                    assert info[4] == 1, list(info)
                    seen2 += 1
                got = run_conv(bk, x, w, scale, shift, BF16, x_pad=(8, 8), y_pad=(16, 8))
                assert np.array_equal(got, y0), (case, float(np.abs(got - y0).max()))
            assert seen2 >= 1, seen2


def case_tube_update(bk, golden):
    """step_tube_update (one refinement step's decode x3 -> cat -> valid_tubes -> frame-index column, utils/utils.py:68-129)
    against the torch restatement in step_amd.tube_math, whose pieces are pinned bit for bit by tube_math_golden.npz."""
    from step_amd import tube_math as TM
    rs = np.random.RandomState(21)
    for N, T, Tw, extend in ((7, 3, 3, 1), (5, 9, 3, 0), (4, 9, 3, 1), (1, 3, 3, 0)):
        xy = rs.uniform(-30, 380, (N, T, 2))
        boxes = np.concatenate([xy, xy + rs.uniform(-2, 160, (N, T, 2))], 2).astype(np.float32)       # some degenerate, some outside
        tubes = np.concatenate([rs.uniform(0, 50, (N, T, 1)).astype(np.float32), boxes], 2)
        loc = (rs.randn(N, T, 4) * 0.2).astype(np.float32)
        floc, lloc = (rs.randn(N, Tw, 4) * 0.2).astype(np.float32), (rs.randn(N, Tw, 4) * 0.2).astype(np.float32)
        clip_of = np.sort(rs.randint(0, 3, N)).astype(np.int32)
        first_off, last_off = 0, T - Tw
        Tn = T + 2 * Tw if extend else T
        d = [bk.dev(a) for a in (tubes, loc, floc, lloc, clip_of)]
        o = [bk.dev(np.full(sh, 7.0, np.float32)) for sh in ((N, T, 4), (N, Tw, 4), (N, Tw, 4), (N, Tn, 5))]
        rc = bk.lib.step_tube_update(d[0].ptr, N, T, d[1].ptr, d[2].ptr, d[3].ptr, Tw, first_off, last_off, d[4].ptr, extend, 400.0, 400.0,
                                     o[0].ptr, o[1].ptr, o[2].ptr, o[3].ptr, bk.stream)
        assert rc == 0, rc
        t = torch.from_numpy
        pl = TM.decode_coef(t(boxes).reshape(-1, 4), t(loc).reshape(-1, 4)).view(N, T, 4)
        pf = TM.decode_coef(t(boxes[:, first_off:first_off + Tw]).reshape(-1, 4), t(floc).reshape(-1, 4)).view(N, Tw, 4)
        pla = TM.decode_coef(t(boxes[:, last_off:last_off + Tw]).reshape(-1, 4), t(lloc).reshape(-1, 4)).view(N, Tw, 4)
        prop = TM.valid_tubes(torch.cat([pf, pl, pla], 1) if extend else pl, 400, 400)
        idx = t(clip_of).float().view(N, 1, 1) * Tn + torch.arange(Tn, dtype=torch.float32).view(1, Tn, 1)
        nxt = torch.cat([idx, prop], 2)
        for got, ref, what in zip(o, (pl, pf, pla, nxt), ("pred_loc", "pred_first", "pred_last", "next")):
            g, r = got.get(), ref.numpy()
            assert g.shape == r.shape
            # exp() is the only operation whose last bit may differ between libraries
            assert np.abs(g - r).max() <= 2e-6 * max(1.0, np.abs(r).max()), (what, N, T, extend, float(np.abs(g - r).max()))
        assert np.array_equal(o[3].get()[..., 0], nxt.numpy()[..., 0])
    assert bk.lib.step_tube_update(None, 0, 3, None, None, None, 3, 0, 0, None, 1, 400.0, 400.0, None, None, None, None, bk.stream) == 0
    assert bk.lib.step_tube_update(None, 2, 3, None, None, None, 4, 0, 0, None, 1, 400.0, 400.0, None, None, None, None, bk.stream) == -2


def case_conv_residual_norelu_f16_and_bias_only(bk, golden):
    rs = np.random.RandomState(11)
    x = rs.randn(2, 32, 1, 7, 7).astype(np.float32)
    w = (rs.randn(64, 32, 1, 1, 1) / 6).astype(np.float32)
    res = rs.randn(2, 64, 1, 7, 7).astype(np.float32)
    got = run_conv(bk, x, w, None, None, F32, relu=True, res=res)
    assert np.abs(got - ref_conv(x, w, None, None, F32, True, res)).max() < 1e-5
    bias = rs.randn(64).astype(np.float32)
    got = run_conv(bk, x, w, None, bias, F16, relu=False)
    ref = ref_conv(x, w, None, bias, F16, False)
    assert np.abs(got - ref).max() / np.abs(ref).max() < tol(F16)


def case_conv_golden_units(bk, golden):
    """Unit3Dpy outputs produced by the reference itself (tests/golden/ops_golden.npz)."""
    g = golden("ops_golden")
    for tag, ci, co, k, shp in (("k3", 20, 24, (3, 3, 3), (2, 20, 3, 6, 7)), ("k1", 20, 12, (1, 1, 1), (2, 20, 3, 6, 7))):
        shapes = {"conv3d.weight": (co, ci) + k}
        for nme in ("weight", "bias", "running_mean", "running_var"):
            shapes["batch3d." + nme] = (co,)
        sd = R.fill_state_dict(shapes, "golden.unit." + tag + ".")
        scale = (sd["batch3d.weight"] / torch.sqrt(sd["batch3d.running_var"] + 1e-5)).numpy()
        shift = (sd["batch3d.bias"] - sd["batch3d.running_mean"] * torch.from_numpy(scale)).numpy()
        x = R.fill_tensor("golden.unit.%s.in" % tag, shp, "image").numpy()
        xp = np.zeros((2, 24) + shp[2:], np.float32)      # Cin 20 -> buffer of 24 channels, conv reads 20 (needs %4)
        xp[:, :20] = x
        got = run_conv(bk, x, sd["conv3d.weight"].numpy(), scale, shift, F32)
        ref = g["unit_%s_out" % tag]
        assert np.abs(got - ref).max() <= 1e-3 * np.abs(ref).max()
        assert np.abs(got - ref).max() <= 2e-5 * np.abs(ref).max()


def case_conv_splitk_few_rows_deep_k(bk, golden):
    """The heads' Linear layers: a handful of rows, K in the thousands, 12 / 60 outputs -> K split over workgroups
    through the caller's workspace; without a workspace the tiled kernel must give the same numbers."""
    rs = np.random.RandomState(21)
    for (rows, Cin, Cout) in ((5, 2056, 12), (70, 2048, 60), (1100, 1024, 12)):   # ragged K (2056 = 128*16 + 8), 1 and 3 row blocks; nine 128-row tiles (> 1024 rows: round 6)
        x = rs.randn(rows, Cin, 1, 1, 1).astype(np.float32)
        w = (rs.randn(Cout, Cin, 1, 1, 1) / np.sqrt(Cin)).astype(np.float32)
        shift = (0.2 * rs.randn(Cout)).astype(np.float32)
        res = rs.randn(rows, Cout, 1, 1, 1).astype(np.float32)
        for dt in (BF16, F16, F32):
            ref = ref_conv(x, w, None, shift, dt, relu=False, res=res)
            got = run_conv(bk, x, w, None, shift, dt, relu=False, res=res, y_pad=(4, 4), use_ws=True)
            if _capi.get_option(bk.lib, "conv_impl") == -1:           # a forced implementation bypasses the planner
                assert run_conv.last_ws_bytes > 0, "split-K path not taken"
            plain = run_conv(bk, x, w, None, shift, dt, relu=False, res=res, y_pad=(4, 4))
            for y in (got, plain):
                err = np.abs(y - ref).max() / np.abs(ref).max()
                assert err < tol(dt), (rows, Cin, Cout, dt, err)
    # shallow layers never ask for a workspace
    d = _capi.ConvDesc(dtype=F32, N=5, D=1, H=1, W=1, Cin=256, Cout=12, kd=1, kh=1, kw=1, x_cstride=256, x_coff=0, y_cstride=12,
                       y_coff=0, res_cstride=0, res_coff=0, relu=0, split=0, y2_cstride=0, y2_coff=0)
    assert bk.lib.step_conv_workspace_bytes(ctypes.byref(d)) == 0


def case_conv_wgrad(bk, golden):
    """Weight gradient kernel against torch autograd of the same conv on identically quantised inputs."""
    rs = np.random.RandomState(33)
    cases = [(2, 24, 40, 3, 5, 19, (3, 3, 3)),      # ragged channels (24 -> one 32-block), 19 pixels per row (16 + 3)
             (1, 72, 100, 2, 6, 7, (1, 3, 3)),      # 2-D kernel, two ci / co tiles
             (1, 40, 70, 1, 9, 130, (1, 1, 1))]     # pointwise: 1170 pixels = full chunks + a ragged tail launch
    # option wgrad_minpix = pixels a wavefront job covers at least (default 512: these small maps become one or two jobs per
    # tile); 64 and 256 force several row-range jobs per tile -- jobs that cross planes and clips, ragged last jobs,
    # pointwise chunks of 64 / 256 pixels with a tail launch
    try:
        for minpix in (64, 256, 0):
            _capi.set_option(bk.lib, "wgrad_minpix", minpix)
            for (N, Cin, Cout, D, H, W, k) in cases:
                x = rs.randn(N, Cin, D, H, W).astype(np.float32)
                gy = rs.randn(N, Cout, D, H, W).astype(np.float32)
                for dt in (F32, BF16):
                    xq = torch.from_numpy(quantize(x, dt)).requires_grad_(False)
                    w = torch.zeros(Cout, Cin, *k, requires_grad=True)
                    F.conv3d(xq, w, padding=tuple(kk // 2 for kk in k)).backward(torch.from_numpy(gy))
                    ref = w.grad.numpy()
                    xpad = np.zeros((N, D, H, W, 8 + Cin), np.float32)
                    xpad[..., 8:] = cl(x)
                    xpad[..., :8] = 1e6                       # channels before the slice: must never be read
                    xd = bk.dev(encode(xpad, dt))
                    gd = bk.dev(np.ascontiguousarray(cl(gy), np.float32))
                    dw = bk.dev(np.full((Cout, Cin) + k, 7.0, np.float32))     # accumulate = 0 must overwrite
                    d = _capi.ConvDesc(dtype=dt, N=N, D=D, H=H, W=W, Cin=Cin, Cout=Cout, kd=k[0], kh=k[1], kw=k[2], x_cstride=8 + Cin, x_coff=8,
                                       y_cstride=Cout, y_coff=0, res_cstride=0, res_coff=0, relu=0, split=0, y2_cstride=0, y2_coff=0)
                    assert bk.lib.step_conv_wgrad(ctypes.byref(d), xd.ptr, gd.ptr, dw.ptr, 0, bk.stream) == 0
                    got = dw.get()
                    err = np.abs(got - ref).max() / np.abs(ref).max()
                    assert err < 2e-5, (N, Cin, Cout, k, dt, minpix, err)
                    assert bk.lib.step_conv_wgrad(ctypes.byref(d), xd.ptr, gd.ptr, dw.ptr, 1, bk.stream) == 0    # accumulate
                    err2 = np.abs(dw.get() - 2 * ref).max() / np.abs(ref).max()
                    assert err2 < 4e-5, err2
                    # the workspace form: every job writes its partial tile, one fixed-order sum -- no atomics, so two runs
                    # are bit-identical; the scratch needs no initialisation (poisoned here)
                    nb = bk.lib.step_conv_wgrad_workspace_bytes(ctypes.byref(d))
                    assert nb > 0 and nb % 16 == 0, (k, nb)
                    outs = []
                    for _ in range(2):
                        ws = bk.dev(np.full(nb // 4, np.nan, np.float32))
                        dw2 = bk.dev(np.full((Cout, Cin) + k, 7.0, np.float32))
                        assert bk.lib.step_conv_wgrad_ws(ctypes.byref(d), xd.ptr, gd.ptr, dw2.ptr, 0, ws.ptr, nb, bk.stream) == 0
                        outs.append(dw2.get())
                    assert np.abs(outs[0] - ref).max() / np.abs(ref).max() < 2e-5, (N, Cin, Cout, k, dt, minpix)
                    assert np.array_equal(outs[0], outs[1])
                    assert bk.lib.step_conv_wgrad_ws(ctypes.byref(d), xd.ptr, gd.ptr, dw2.ptr, 1, ws.ptr, nb, bk.stream) == 0
                    assert np.abs(dw2.get() - 2 * ref).max() / np.abs(ref).max() < 4e-5
                    assert bk.lib.step_conv_wgrad_ws(ctypes.byref(d), xd.ptr, gd.ptr, dw2.ptr, 0, ws.ptr, nb - 16, bk.stream) < 0   # short scratch refused
                    assert bk.lib.step_conv_wgrad_ws(ctypes.byref(d), xd.ptr, gd.ptr, dw2.ptr, 0, None, 0, bk.stream) == 0          # none: the atomics form
                    assert np.abs(dw2.get() - ref).max() / np.abs(ref).max() < 2e-5
    finally:
        _capi.set_option(bk.lib, "wgrad_minpix", 0)

    d = _capi.ConvDesc(dtype=F32, N=1, D=1, H=4, W=4, Cin=8, Cout=8, kd=1, kh=2, kw=2, x_cstride=8, x_coff=0, y_cstride=8, y_coff=0,
                       res_cstride=0, res_coff=0, relu=0, split=0, y2_cstride=0, y2_coff=0)
    assert bk.lib.step_conv_wgrad(ctypes.byref(d), None, None, None, 0, bk.stream) < 0     # even kernels: unsupported / null


def case_conv_wgrad16(bk, golden):
    """step_conv_wgrad16 (weight gradient on the 16-bit matrix instructions: x AND dy in bf16 / fp16, fp32 accumulation)
    against torch autograd on identically quantised operands: 16-bit x 16-bit products are exact in fp32, so only the summation
    order differs (2e-5)."""
    rs = np.random.RandomState(35)
    cases = [(2, 24, 40, 3, 5, 19, (3, 3, 3)), (1, 72, 100, 2, 6, 7, (1, 3, 3)), (1, 40, 70, 1, 9, 130, (1, 1, 1))]
    cases += [(1, 64, 72, 3, 20, 14, (3, 3, 3)),     # two row chunks per plane (20 rows of 14 > 224 pixels), a ragged second co tile
              (2, 16, 32, 1, 3, 100, (3, 3, 3)),     # 100-pixel rows: the wide-halo instantiation (two rows per chunk, then a ragged one), a single ci block
              (1, 24, 16, 2, 7, 56, (3, 3, 3)),      # 56-pixel rows: wide halo too (chunks of four and three rows instead of 3 + 2 + 2)
              (1, 392, 72, 2, 9, 40, (1, 1, 1)),     # pointwise through the LDS-tiled form (Cin >= 384, < 2048 pixels): 720 pixels = 5 chunks + a ragged one, three ci tiles of 192 (the last one ragged)
              (1, 72, 40, 2, 36, 40, (1, 1, 1)),     # pointwise as a pixel stream (>= 2048 pixels): one 64 x 128 tile with ragged channels, 90 stages
              (1, 392, 136, 1, 30, 70, (1, 1, 1))]   # ... 2 x 2 tiles of 96 x 256 (both axes ragged), 2100 pixels = 65 stages + 20 pixels
    try:
        # 64 with wgrad16_lds = 0: the per-tap 16-bit form with several row-range jobs per tile; None: the LDS-tiled form
        # (wgrad16_lds = 2: the pointwise layers on the LDS-tiled / per-tap forms instead of the pixel stream)
        for minpix, ldsopt in ((64, 0), (None, 1), (None, 2)):
            _capi.set_option(bk.lib, "wgrad_minpix", minpix or 0)
            _capi.set_option(bk.lib, "wgrad16_lds", ldsopt)
            for (N, Cin, Cout, D, H, W, k) in cases:
                x = rs.randn(N, Cin, D, H, W).astype(np.float32)
                gy = rs.randn(N, Cout, D, H, W).astype(np.float32)
                for dt in (BF16, F16):
                    w = torch.zeros(Cout, Cin, *k, requires_grad=True)
                    F.conv3d(torch.from_numpy(quantize(x, dt)), w, padding=tuple(kk // 2 for kk in k)).backward(torch.from_numpy(quantize(gy, dt)))
                    ref = w.grad.numpy()
                    xpad = np.zeros((N, D, H, W, 8 + Cin), np.float32)
                    xpad[..., 8:] = cl(x)
                    xpad[..., :8] = 1e4
                    xd = bk.dev(encode(xpad, dt))
                    gd = bk.dev(encode(cl(gy), dt))
                    dw = bk.dev(np.full((Cout, Cin) + k, 7.0, np.float32))
                    d = _capi.ConvDesc(dtype=dt, N=N, D=D, H=H, W=W, Cin=Cin, Cout=Cout, kd=k[0], kh=k[1], kw=k[2], x_cstride=8 + Cin, x_coff=8,
                                       y_cstride=Cout, y_coff=0, res_cstride=0, res_coff=0, relu=0, split=0, y2_cstride=0, y2_coff=0)
                    assert bk.lib.step_conv_wgrad16(ctypes.byref(d), xd.ptr, gd.ptr, dw.ptr, 0, bk.stream) == 0
                    err = np.abs(dw.get() - ref).max() / np.abs(ref).max()
                    assert err < 2e-5, (N, Cin, Cout, k, dt, minpix, err)
                    assert bk.lib.step_conv_wgrad16(ctypes.byref(d), xd.ptr, gd.ptr, dw.ptr, 1, bk.stream) == 0
                    assert np.abs(dw.get() - 2 * ref).max() / np.abs(ref).max() < 4e-5
                    # the workspace form (LDS-tiled kernel + fixed-order sum of partial tiles): same result, and bit-reproducible
                    nb = bk.lib.step_conv_wgrad16_workspace_bytes(ctypes.byref(d))
                    if minpix is None and (k[1] == 3 or N * D * H * W >= 2048 or (N * D * H * W >= 512 and Cin >= 384)) and Cin % 8 == 0 and Cout % 8 == 0:
                        assert nb > 0 and nb % 16 == 0, (k, nb)
                    if nb:
                        outs = []
                        for _ in range(2):
                            ws = bk.dev(np.full(nb // 4, np.nan, np.float32))      # scratch needs no initialisation: poison it
                            dw2 = bk.dev(np.full((Cout, Cin) + k, 7.0, np.float32))
                            assert bk.lib.step_conv_wgrad16_ws(ctypes.byref(d), xd.ptr, gd.ptr, dw2.ptr, 0, ws.ptr, nb, bk.stream) == 0
                            outs.append(dw2.get())
                        assert np.abs(outs[0] - ref).max() / np.abs(ref).max() < 2e-5, (N, Cin, Cout, k, dt)
                        assert np.array_equal(outs[0], outs[1])
                        assert bk.lib.step_conv_wgrad16_ws(ctypes.byref(d), xd.ptr, gd.ptr, dw2.ptr, 1, ws.ptr, nb, bk.stream) == 0
                        assert np.abs(dw2.get() - 2 * ref).max() / np.abs(ref).max() < 4e-5
    finally:
        _capi.set_option(bk.lib, "wgrad_minpix", 0)
        _capi.set_option(bk.lib, "wgrad16_lds", 1)
    d = _capi.ConvDesc(dtype=F32, N=1, D=1, H=4, W=4, Cin=8, Cout=8, kd=1, kh=1, kw=1, x_cstride=8, x_coff=0, y_cstride=8, y_coff=0,
                       res_cstride=0, res_coff=0, relu=0, split=0, y2_cstride=0, y2_coff=0)
    z = bk.dev(np.zeros(64, np.float32))
    assert bk.lib.step_conv_wgrad16(ctypes.byref(d), z.ptr, z.ptr, z.ptr, 0, bk.stream) == -4       # fp32 storage: unsupported


def case_wgrad_partial_and_grouped_reduce(bk, golden):
    """step_conv_wgrad_partial + step_wgrad_reduce_group (the fixed-order sums of several layers' partial tiles as ONE launch: an
    Inception block's six weight gradients) give BIT-IDENTICAL gradients to step_conv_wgrad_ws / step_conv_wgrad16_ws layer by layer:
    a 3x3x3 and a deep pointwise layer on the LDS-tiled 16-bit form, a shallow pointwise one on the per-tap 16-bit form, an fp32 layer,
    accumulate on and off, the gradient read in place as a channel SLICE of a wider buffer (what _MixedTrainFn hands over)."""
    rs = np.random.RandomState(36)
    layers = [(BF16, True, 1, 64, 72, 3, 20, 14, (3, 3, 3), 0), (BF16, True, 1, 392, 72, 2, 9, 40, (1, 1, 1), 1), (BF16, True, 2, 40, 24, 1, 9, 30, (1, 1, 1), 0),
              (F32, False, 1, 24, 40, 2, 6, 7, (3, 3, 3), 1), (F16, True, 1, 16, 32, 1, 3, 100, (3, 3, 3), 0),
              (BF16, True, 1, 72, 40, 2, 36, 40, (1, 1, 1), 1)]              # a pointwise layer on the pixel-stream form (dense per-slice images)
    items = (_capi.WgradReduceItem * len(layers))()
    keep, expect, got = [], [], []
    for li, (dt, w16, N, Cin, Cout, D, H, W, k, acc) in enumerate(layers):
        x = rs.randn(N, D, H, W, Cin).astype(np.float32)
        wide = rs.randn(N, D, H, W, Cout + 16).astype(np.float32)              # the gradient is channels [8, 8 + Cout) of this buffer
        gdt = dt if w16 else F32
        xd, gwide = bk.dev(encode(x, dt)), bk.dev(encode(wide, gdt))
        gdense = bk.dev(encode(np.ascontiguousarray(wide[..., 8:8 + Cout]), gdt))
        es = 2 if w16 else 4
        gslice_ptr = (gwide.ptr + 8 * es) if isinstance(gwide.ptr, int) else ctypes.c_void_p(gwide.ptr.value + 8 * es)
        init = rs.randn(Cout, Cin, *k).astype(np.float32)
        d = _capi.ConvDesc(dtype=dt, N=N, D=D, H=H, W=W, Cin=Cin, Cout=Cout, kd=k[0], kh=k[1], kw=k[2], x_cstride=Cin, x_coff=0,
                           y_cstride=Cout, y_coff=0, res_cstride=0, res_coff=0, relu=0, split=0, y2_cstride=0, y2_coff=0)
        nb = (bk.lib.step_conv_wgrad16_workspace_bytes if w16 else bk.lib.step_conv_wgrad_workspace_bytes)(ctypes.byref(d))
        assert nb > 0
        ws = bk.dev(np.full(nb // 4, np.nan, np.float32))
        dw = bk.dev(init.copy())
        fn = bk.lib.step_conv_wgrad16_ws if w16 else bk.lib.step_conv_wgrad_ws
        assert fn(ctypes.byref(d), xd.ptr, gdense.ptr, dw.ptr, acc, ws.ptr, nb, bk.stream) == 0
        expect.append(dw.get().copy())
        ds = _capi.ConvDesc(dtype=dt, N=N, D=D, H=H, W=W, Cin=Cin, Cout=Cout, kd=k[0], kh=k[1], kw=k[2], x_cstride=Cin, x_coff=0,
                            y_cstride=Cout + 16, y_coff=0, res_cstride=0, res_coff=0, relu=0, split=0, y2_cstride=0, y2_coff=0)
        ws2 = bk.dev(np.full(nb // 4, np.nan, np.float32))
        dw2 = bk.dev(init.copy())
        assert bk.lib.step_conv_wgrad_partial(ctypes.byref(ds), xd.ptr, gslice_ptr, int(w16), dw2.ptr, acc, ws2.ptr, nb, ctypes.byref(items[li]), bk.stream) == 0
        assert items[li].kind in (1, 2, 3)
        keep.append((ws2, xd, gwide))
        got.append(dw2)
    assert bk.lib.step_wgrad_reduce_group(items, len(layers), bk.stream) == 0
    for li in range(len(layers)):
        assert np.array_equal(got[li].get(), expect[li]), layers[li]
    assert bk.lib.step_wgrad_reduce_group(items, 9, bk.stream) == -2 and bk.lib.step_wgrad_reduce_group(items, 0, bk.stream) == 0
    none = _capi.WgradReduceItem()
    assert bk.lib.step_wgrad_reduce_group(ctypes.byref(none), 1, bk.stream) == 0          # kind 0: nothing pending
    # the SAME gradient twice in one group (a unit used twice in a graph; leftovers of an earlier backward): the members of one launch run
    # concurrently and each ends in a plain read-modify-write, so the entry point must order them as separate launches -- the result
    # equals two accumulating single-layer calls
    dt, N, Cin, Cout, D, H, W, k = BF16, 1, 64, 72, 2, 10, 14, (3, 3, 3)
    d = _capi.ConvDesc(dtype=dt, N=N, D=D, H=H, W=W, Cin=Cin, Cout=Cout, kd=k[0], kh=k[1], kw=k[2], x_cstride=Cin, x_coff=0,
                       y_cstride=Cout, y_coff=0, res_cstride=0, res_coff=0, relu=0, split=0, y2_cstride=0, y2_coff=0)
    nb = bk.lib.step_conv_wgrad16_workspace_bytes(ctypes.byref(d))
    xs = [bk.dev(encode(rs.randn(N, D, H, W, Cin).astype(np.float32), dt)) for _ in range(2)]
    gs = [bk.dev(encode(rs.randn(N, D, H, W, Cout).astype(np.float32), dt)) for _ in range(2)]
    init = rs.randn(Cout, Cin, *k).astype(np.float32)
    dwa, dwb = bk.dev(init.copy()), bk.dev(init.copy())
    wsa = [bk.dev(np.full(nb // 4, np.nan, np.float32)) for _ in range(2)]
    for i in range(2):
        assert bk.lib.step_conv_wgrad16_ws(ctypes.byref(d), xs[i].ptr, gs[i].ptr, dwa.ptr, 1, wsa[i].ptr, nb, bk.stream) == 0
    two = (_capi.WgradReduceItem * 2)()
    wsb = [bk.dev(np.full(nb // 4, np.nan, np.float32)) for _ in range(2)]
    for i in range(2):
        assert bk.lib.step_conv_wgrad_partial(ctypes.byref(d), xs[i].ptr, gs[i].ptr, 1, dwb.ptr, 1, wsb[i].ptr, nb, ctypes.byref(two[i]), bk.stream) == 0
    assert two[0].dw == two[1].dw
    assert bk.lib.step_wgrad_reduce_group(two, 2, bk.stream) == 0
    assert np.array_equal(dwb.get(), dwa.get())


def case_stem_pool_fused(bk, golden):
    """step_stem_pool_forward (the stem with maxPool3d_2a taken on its tiles, models/i3dpt.py:186-196) is BIT-IDENTICAL to step_stem_forward
    followed by step_maxpool3d_tf (1,3,3) / (1,2,2): maps of several tiles in both directions with partial last tiles (seams completed by
    the second launch, corners by exactly one thread), odd sizes (ceil-mode windows that hang over the padded border), a single-tile map
    (no seam launch), a channel-slice destination; unsupported shapes are refused so that the caller falls back."""
    rs = np.random.RandomState(21)
    Cout = 64
    w = (rs.randn(Cout, 3, 7, 7, 7) / np.sqrt(1029)).astype(np.float32)
    scale = (1 + 0.1 * rs.randn(Cout)).astype(np.float32)
    shift = (0.2 * rs.randn(Cout)).astype(np.float32)
    for (N, T, H, W), dt, ycs, yoff in (((1, 4, 72, 80), BF16, 64, 0), ((2, 2, 50, 52), F16, 64, 0), ((1, 2, 20, 24), BF16, 80, 8), ((1, 2, 34, 68), BF16, 64, 0)):
        x = rs.uniform(-1, 1, (N, T, 3, H, W)).astype(np.float32)
        To, Ho, Wo = (T - 2) // 2 + 1, (H - 2) // 2 + 1, (W - 2) // 2 + 1
        Hp, Wp = bk.lib.step_pool_out_size(Ho, 3, 2), bk.lib.step_pool_out_size(Wo, 3, 2)
        wp = bk.dev(np.zeros(bk.lib.step_stem_packed_elems(Cout), NP_DT[dt]))
        wd = bk.dev(np.ascontiguousarray(w, np.float32))
        assert bk.lib.step_stem_pack_weight(wd.ptr, Cout, dt, wp.ptr, bk.stream) == 0
        xe, sc, sh = bk.dev(encode(x, dt)), bk.dev(scale), bk.dev(shift)
        y = bk.dev(np.zeros((N, To, Ho, Wo, Cout), NP_DT[dt]))
        assert bk.lib.step_stem_forward(dt, xe.ptr, N, T, H, W, wp.ptr, sc.ptr, sh.ptr, 1, Cout, y.ptr, Cout, 0, bk.stream) == 0
        ref = bk.dev(np.zeros((N, To, Hp, Wp, Cout), NP_DT[dt]))
        assert bk.lib.step_maxpool3d_tf(dt, y.ptr, N, To, Ho, Wo, Cout, Cout, 0, 1, 3, 3, 1, 2, 2, ref.ptr, Cout, 0, bk.stream) == 0
        wsb = bk.lib.step_stem_pool_workspace_bytes(dt, N, T, H, W, Cout)
        assert wsb > 0 and wsb % 16 == 0
        ws = bk.dev(np.full(wsb // 2, 0x7f7f, np.uint16))                      # a huge positive pattern: a seam read from a slot no tile wrote would win the max
        out = bk.dev(np.full((N, To, Hp, Wp, ycs), 0x3c00 if dt == F16 else 0x3f80, NP_DT[dt] if dt == F16 else np.uint16).view(NP_DT[dt]))
        before = out.get().copy()
        assert bk.lib.step_stem_pool_forward(dt, xe.ptr, N, T, H, W, wp.ptr, sc.ptr, sh.ptr, Cout, out.ptr, ycs, yoff, ws.ptr, wsb, bk.stream) == 0
        got = out.get()
        assert np.array_equal(got[..., yoff:yoff + Cout].view(np.uint16), ref.get().view(np.uint16)), (N, T, H, W)
        if ycs > Cout:                                                          # the rest of the wider buffer is untouched
            keep = np.ones(ycs, bool)
            keep[yoff:yoff + Cout] = False
            assert np.array_equal(got[..., keep].view(np.uint16), before[..., keep].view(np.uint16))
        assert bk.lib.step_stem_pool_forward(dt, xe.ptr, N, T, H, W, wp.ptr, sc.ptr, sh.ptr, Cout, out.ptr, ycs, yoff, ws.ptr, wsb - 16, bk.stream) < 0
    assert bk.lib.step_stem_pool_workspace_bytes(F32, 1, 4, 72, 80, 64) == 0 and bk.lib.step_stem_pool_workspace_bytes(BF16, 1, 4, 72, 82, 64) == 0 \
        and bk.lib.step_stem_pool_workspace_bytes(BF16, 1, 4, 72, 80, 32) == 0
    assert bk.lib.step_stem_pool_forward(F32, xe.ptr, 1, 4, 72, 80, wp.ptr, sc.ptr, sh.ptr, 64, out.ptr, 64, 0, ws.ptr, wsb, bk.stream) == -4


def case_stem_pool_forward_u8(bk, golden):
    """step_stem_pool_forward_u8 (the stem stages the decoder's uint8 frames [N,T,H,W,3] itself: normalisation + storage rounding through a
    table built in the prologue) is BIT-IDENTICAL to step_clip_from_u8 (into the storage type) + step_stem_pool_forward: all three scale
    modes, per-channel mean / std, both 16-bit types, maps of several tiles with partial tiles and image borders (out-of-image quads are
    zeros AFTER normalisation, as the padded clip's), byte extremes 0 / 255 on the border; misaligned frames are refused."""
    rs = np.random.RandomState(23)
    Cout = 64
    w = (rs.randn(Cout, 3, 7, 7, 7) / np.sqrt(1029)).astype(np.float32)
    scale = (1 + 0.1 * rs.randn(Cout)).astype(np.float32)
    shift = (0.2 * rs.randn(Cout)).astype(np.float32)
    mean, std = np.array([0.1, -0.2, 0.3], np.float32), np.array([1.0, 0.5, 2.0], np.float32)
    for (N, T, H, W), dt, mode, norm in (((1, 4, 40, 72), BF16, 2, False), ((2, 3, 34, 36), F16, 1, True), ((1, 2, 20, 24), BF16, 0, True)):
        fr = rs.randint(0, 256, (N, T, H, W, 3)).astype(np.uint8)
        fr[:, :, 0, :4] = 0
        fr[:, :, -1, -4:] = 255
        To, Ho, Wo = (T - 2) // 2 + 1, (H - 2) // 2 + 1, (W - 2) // 2 + 1
        Hp, Wp = bk.lib.step_pool_out_size(Ho, 3, 2), bk.lib.step_pool_out_size(Wo, 3, 2)
        wp = bk.dev(np.zeros(bk.lib.step_stem_packed_elems(Cout), NP_DT[dt]))
        wd = bk.dev(np.ascontiguousarray(w, np.float32))
        assert bk.lib.step_stem_pack_weight(wd.ptr, Cout, dt, wp.ptr, bk.stream) == 0
        sc, sh, frd = bk.dev(scale), bk.dev(shift), bk.dev(fr)
        m = (ctypes.c_float * 3)(*mean.tolist()) if norm else None
        sd = (ctypes.c_float * 3)(*std.tolist()) if norm else None
        clip = bk.dev(np.zeros((N, T, 3, H, W), NP_DT[dt]))
        assert bk.lib.step_clip_from_u8(frd.ptr, N, T, H, W, mode, m, sd, dt, clip.ptr, bk.stream) == 0
        wsb = bk.lib.step_stem_pool_workspace_bytes(dt, N, T, H, W, Cout)
        assert wsb > 0
        ws = bk.dev(np.full(wsb // 2, 0x7f7f, np.uint16))
        want = bk.dev(np.zeros((N, To, Hp, Wp, Cout), NP_DT[dt]))
        assert bk.lib.step_stem_pool_forward(dt, clip.ptr, N, T, H, W, wp.ptr, sc.ptr, sh.ptr, Cout, want.ptr, Cout, 0, ws.ptr, wsb, bk.stream) == 0
        ws2 = bk.dev(np.full(wsb // 2, 0x7f7f, np.uint16))
        got = bk.dev(np.zeros((N, To, Hp, Wp, Cout), NP_DT[dt]))
        assert bk.lib.step_stem_pool_forward_u8(dt, frd.ptr, N, T, H, W, mode, m, sd, wp.ptr, sc.ptr, sh.ptr, Cout, got.ptr, Cout, 0, ws2.ptr, wsb,
                                                bk.stream) == 0
        g_, w_ = got.get().view(np.uint16), want.get().view(np.uint16)
        assert np.array_equal(g_, w_), ((N, T, H, W), dt, mode, int((g_ != w_).sum()))
        assert np.abs(decode(got.get(), dt)).max() > 0
    assert bk.lib.step_stem_pool_forward_u8(F32, frd.ptr, N, T, H, W, 2, None, None, wp.ptr, sc.ptr, sh.ptr, Cout, got.ptr, Cout, 0, ws2.ptr, wsb, bk.stream) == -4
    assert bk.lib.step_stem_pool_forward_u8(dt, frd.ptr, N, T, H, W, 3, None, None, wp.ptr, sc.ptr, sh.ptr, Cout, got.ptr, Cout, 0, ws2.ptr, wsb, bk.stream) == -2
    off1 = (frd.ptr + 1) if isinstance(frd.ptr, int) else ctypes.c_void_p(frd.ptr.value + 1)
    assert bk.lib.step_stem_pool_forward_u8(dt, off1, N, T, H, W, 2, None, None, wp.ptr, sc.ptr, sh.ptr, Cout, got.ptr, Cout, 0, ws2.ptr, wsb, bk.stream) == -5


def case_stem_wgrad(bk, golden):
    rs = np.random.RandomState(44)
    N, T, H, W, Cout = 1, 6, 21, 70, 40                     # Ho = 10 (two row chunks), Wo = 35 (edge, interior and edge steps); Cout not a multiple of 32
    x = rs.randn(N, T, 3, H, W).astype(np.float32)
    To, Ho, Wo = (T - 2) // 2 + 1, (H - 2) // 2 + 1, (W - 2) // 2 + 1
    gy = rs.randn(N, Cout, To, Ho, Wo).astype(np.float32)
    for dt in (F32, BF16):
        xq = torch.from_numpy(quantize(x, dt))
        w = torch.zeros(Cout, 3, 7, 7, 7, requires_grad=True)
        xp = F.pad(xq.permute(0, 2, 1, 3, 4), (2, 3, 2, 3, 2, 3))
        y = F.conv3d(xp, w, stride=2)
        assert tuple(y.shape[2:]) == (To, Ho, Wo)
        y.backward(torch.from_numpy(gy))
        ref = w.grad.numpy()
        xd = bk.dev(encode(x, dt))
        gd = bk.dev(np.ascontiguousarray(cl(gy), np.float32))
        dw = bk.dev(np.full((Cout, 3, 7, 7, 7), -3.0, np.float32))
        assert bk.lib.step_stem_wgrad(dt, xd.ptr, N, T, H, W, gd.ptr, Cout, dw.ptr, 0, bk.stream) == 0
        err = np.abs(dw.get() - ref).max() / np.abs(ref).max()
        assert err < 2e-5, (dt, err)
        # the workspace form: partial tiles + fixed-order sum (no atomics): same values, run-to-run bit identity, accumulate
        nb = bk.lib.step_stem_wgrad_workspace_bytes(N, T, H, W, Cout)
        assert nb > 0 and nb % 16 == 0
        outs = []
        for _ in range(2):
            ws = bk.dev(np.full(nb // 4, np.nan, np.float32))
            dw2 = bk.dev(np.full((Cout, 3, 7, 7, 7), -3.0, np.float32))
            assert bk.lib.step_stem_wgrad_ws(dt, xd.ptr, N, T, H, W, gd.ptr, Cout, dw2.ptr, 0, ws.ptr, nb, bk.stream) == 0
            outs.append(dw2.get())
        assert np.abs(outs[0] - ref).max() / np.abs(ref).max() < 2e-5
        assert np.array_equal(outs[0], outs[1])
        assert bk.lib.step_stem_wgrad_ws(dt, xd.ptr, N, T, H, W, gd.ptr, Cout, dw2.ptr, 1, ws.ptr, nb, bk.stream) == 0
        assert np.abs(dw2.get() - 2 * ref).max() / np.abs(ref).max() < 4e-5
        assert bk.lib.step_stem_wgrad_ws(dt, xd.ptr, N, T, H, W, gd.ptr, Cout, dw2.ptr, 0, ws.ptr, nb - 16, bk.stream) < 0
    assert bk.lib.step_stem_wgrad_workspace_bytes(0, T, H, W, Cout) == 0


def case_stem_wgrad16(bk, golden):
    """step_stem_wgrad16 (16-bit clip and gradient, fp32 accumulation, workspace + fixed-order sum) against torch's conv3d
    weight gradient of the same quantized operands: ragged row / column tiles, two column tiles, a partial channel block,
    accumulate, run-to-run bit identity, and the shapes it hands back to step_stem_wgrad."""
    rs = np.random.RandomState(45)
    for (N, T, H, W, Cout) in ((2, 6, 22, 72, 40), (1, 4, 8, 240, 32), (1, 5, 13, 16, 64)):
        x = rs.randn(N, T, 3, H, W).astype(np.float32)
        To, Ho, Wo = (T - 2) // 2 + 1, (H - 2) // 2 + 1, (W - 2) // 2 + 1
        gy = rs.randn(N, Cout, To, Ho, Wo).astype(np.float32)
        for dt in (BF16, F16):
            xq, gq = torch.from_numpy(quantize(x, dt)), torch.from_numpy(quantize(gy, dt))
            w = torch.zeros(Cout, 3, 7, 7, 7, requires_grad=True, dtype=torch.float64)
            y = F.conv3d(F.pad(xq.permute(0, 2, 1, 3, 4).double(), (2, 3, 2, 3, 2, 3)), w, stride=2)
            assert tuple(y.shape[2:]) == (To, Ho, Wo)
            y.backward(gq.double())
            ref = w.grad.numpy()
            xd, gd = bk.dev(encode(x, dt)), bk.dev(encode(np.ascontiguousarray(cl(gy)), dt))
            wsb = bk.lib.step_stem_wgrad16_workspace_bytes(dt, N, T, H, W, Cout)
            assert wsb > 0
            ws = bk.dev(np.full(wsb // 4, 5.0, np.float32))
            dw = bk.dev(np.full((Cout, 3, 7, 7, 7), -3.0, np.float32))
            assert bk.lib.step_stem_wgrad16(dt, xd.ptr, N, T, H, W, gd.ptr, Cout, dw.ptr, 0, ws.ptr, wsb, bk.stream) == 0
            got = dw.get().copy()
            err = np.abs(got - ref).max() / np.abs(ref).max()
            assert err < 2e-5, (N, T, H, W, Cout, dt, err)
            assert bk.lib.step_stem_wgrad16(dt, xd.ptr, N, T, H, W, gd.ptr, Cout, dw.ptr, 1, ws.ptr, wsb, bk.stream) == 0
            assert np.abs(dw.get() - 2 * got).max() <= 1e-6 * np.abs(got).max()
            dw2 = bk.dev(np.zeros((Cout, 3, 7, 7, 7), np.float32))
            assert bk.lib.step_stem_wgrad16(dt, xd.ptr, N, T, H, W, gd.ptr, Cout, dw2.ptr, 0, ws.ptr, wsb, bk.stream) == 0
            assert np.array_equal(dw2.get(), got)                                   # deterministic: no atomics
            assert bk.lib.step_stem_wgrad16(dt, xd.ptr, N, T, H, W, gd.ptr, Cout, dw2.ptr, 0, ws.ptr, wsb - 16, bk.stream) < 0
    # an empty batch: no launch, the gradient is cleared (or left alone when accumulating)
    dz = bk.dev(np.full((64, 3, 7, 7, 7), 2.0, np.float32))
    assert bk.lib.step_stem_wgrad16(BF16, None, 0, 6, 24, 72, None, 64, dz.ptr, 1, None, 0, bk.stream) == 0 and float(dz.get().min()) == 2.0
    assert bk.lib.step_stem_wgrad16(BF16, None, 0, 6, 24, 72, None, 64, dz.ptr, 0, None, 0, bk.stream) == 0 and not dz.get().any()
    assert bk.lib.step_stem_wgrad16_workspace_bytes(F32, 1, 6, 22, 72, 64) == 0       # fp32, W % 8, Cout % 8: step_stem_wgrad's
    assert bk.lib.step_stem_wgrad16_workspace_bytes(BF16, 1, 6, 22, 70, 64) == 0
    assert bk.lib.step_stem_wgrad16_workspace_bytes(BF16, 1, 6, 22, 72, 60) == 0
    assert bk.lib.step_stem_wgrad16(BF16, xd.ptr, 1, 6, 22, 70, gd.ptr, 64, dw.ptr, 0, ws.ptr, wsb, bk.stream) < 0


def case_conv_split_two_destinations(bk, golden):
    """1x1x1 conv whose output channels [0,split) and [split,Cout) land in two different buffers."""
    rs = np.random.RandomState(13)
    N, Cin, D, H, W = 2, 48, 2, 5, 7
    co_a, co_b = 40, 56                      # split inside a 32-channel block on purpose
    x = rs.randn(N, Cin, D, H, W).astype(np.float32)
    w = (rs.randn(co_a + co_b, Cin, 1, 1, 1) / 7).astype(np.float32)
    scale = (1 + 0.1 * rs.randn(co_a + co_b)).astype(np.float32)
    shift = (0.2 * rs.randn(co_a + co_b)).astype(np.float32)
    for dt in (F32, BF16):
        ref = ref_conv(x, w, scale, shift, dt)
        xe = bk.dev(encode(cl(x), dt))
        wp = pack_weight(bk, w, dt)
        ya = bk.dev(np.zeros((N, D, H, W, 64), NP_DT[dt]))     # slice [8, 48) of a 64-wide buffer
        yb = bk.dev(np.zeros((N, D, H, W, 72), NP_DT[dt]))     # slice [16, 72)
        d = _capi.ConvDesc(dtype=dt, N=N, D=D, H=H, W=W, Cin=Cin, Cout=co_a + co_b, kd=1, kh=1, kw=1, x_cstride=Cin, x_coff=0,
                           y_cstride=64, y_coff=8, res_cstride=0, res_coff=0, relu=1, split=co_a, y2_cstride=72, y2_coff=16)
        sc, sh = bk.dev(scale), bk.dev(shift)
        assert bk.lib.step_conv_forward(ctypes.byref(d), xe.ptr, wp.ptr, sc.ptr, sh.ptr, None, ya.ptr, yb.ptr, bk.stream) == 0
        a, b = decode(ya.get(), dt), decode(yb.get(), dt)
        assert not a[..., :8].any() and not a[..., 48:].any() and not b[..., :16].any()
        got = np.concatenate([uncl(a[..., 8:48]), uncl(b[..., 16:72])], 1)
        assert np.abs(got - ref).max() / np.abs(ref).max() < tol(dt)
    # a 3x3x3 conv refuses a split
    d = _capi.ConvDesc(dtype=F32, N=1, D=1, H=4, W=4, Cin=8, Cout=64, kd=3, kh=3, kw=3, x_cstride=8, x_coff=0, y_cstride=32,
                       y_coff=0, res_cstride=0, res_coff=0, relu=1, split=32, y2_cstride=32, y2_coff=0)
    assert bk.lib.step_conv_forward(ctypes.byref(d), xe.ptr, wp.ptr, None, None, None, ya.ptr, yb.ptr, bk.stream) == -4


def case_conv_general_box_row_groups(bk, golden):
    """General boxes whose width is just below a multiple of 16 (the 14- and 28-wide C2 maps, 13): the accumulator rows are
    assigned so that every 16-lane LDS read group is one run of columns of one box row (p.gmode = 1).  Same pixels, same
    per-pixel accumulation order: bit-identical to the linear walk (option conv_gmode = 0) and within tolerance of the oracle;
    the plan says which mode ran."""
    rs = np.random.RandomState(41)
    # (N, Cin, Cout, D, H, W, k)
    cases = ((1, 64, 64, 8, 14, 14, (3, 3, 3)), (1, 64, 40, 4, 9, 28, (3, 3, 3)), (1, 64, 64, 8, 12, 28, (3, 3, 3)),
             (2, 64, 96, 6, 14, 14, (3, 3, 3)), (1, 64, 64, 2, 10, 30, (3, 3, 3)))
    _capi.set_option(bk.lib, "conv_waves", 8)                 # (the 256-pixel tiles: boxes of 14 x 14, 2 x 4 x 28, ...)
    grouped = 0
    try:
        for (N, Cin, Cout, D, H, W, k) in cases:
            x = rs.randn(N, Cin, D, H, W).astype(np.float32)
            w = (rs.randn(Cout, Cin, *k) / np.sqrt(Cin * k[0] * 9)).astype(np.float32)
            scale = (1 + 0.1 * rs.randn(Cout)).astype(np.float32)
            shift = (0.2 * rs.randn(Cout)).astype(np.float32)
            for dt in (BF16, F32):
                ref = ref_conv(x, w, scale, shift, dt)
                outs = {}
                for mode in ("1", "0"):
                    _capi.set_option(bk.lib, "conv_gmode", int(mode))
                    d = _capi.ConvDesc(dtype=dt, N=N, D=D, H=H, W=W, Cin=Cin, Cout=Cout, kd=k[0], kh=k[1], kw=k[2], x_cstride=Cin, x_coff=0,
                                       y_cstride=Cout, y_coff=0, res_cstride=0, res_coff=0, relu=1, split=0, y2_cstride=0, y2_coff=0)
                    info = (ctypes.c_int * 10)()
                    assert bk.lib.step_conv_plan_info(ctypes.byref(d), info, 10) == 0
                    assert info[0] == 1, list(info)                                         # the pipelined kernel
                    fits = info[5] * info[6] * ((info[7] + 15) // 16) <= 2 * info[3]         # box rows x 16-column runs <= 16-lane slots of the tile
                    assert info[8] == int(mode == "1" and info[1] == 0 and (info[7] & 15) >= 12 and fits), (mode, list(info))
                    grouped += info[8]
                    outs[mode] = run_conv(bk, x, w, scale, shift, dt)
                    assert np.abs(outs[mode] - ref).max() / np.abs(ref).max() < tol(dt), (mode, W, dt)
                assert np.array_equal(outs["1"], outs["0"]), (W, dt)
        assert grouped >= 4, grouped                                                      # (the mode really ran)
    finally:
        _capi.set_option(bk.lib, "conv_gmode", 1)
        _capi.set_option(bk.lib, "conv_waves", 0)


def case_conv_group_matches_separate_launches(bk, golden):
    """step_conv_forward_group: two independent 3x3x3 convs (an Inception block's branch_1 / branch_2 shapes: different inputs of one
    scratch buffer, different depths, outputs = channel slices of one buffer) as ONE grid -- bit-identical to two step_conv_forward
    calls; a narrow member inside a deeper instantiation leaves a wave group without channel blocks (it must skip, not store); fp32 /
    pointwise members are launched separately with the same results."""
    rs = np.random.RandomState(51)
    buf = ctypes.create_string_buffer(256)
    G14, H7 = (1, 8, 14, 14), (6, 3, 7, 7)                     # general 8 x 2 x 14 boxes (7 pixel tiles) | the heads' maps: 4-plane 8x8 tiles
    for dt, (N, D, H, W), (ci0, co0, ci1, co1), merged in ((BF16, G14, (64, 200, 64, 40), True), (F16, G14, (96, 64, 64, 64), True),
                                                           (BF16, G14, (64, 96, 64, 160), True), (BF16, H7, (160, 320, 64, 128), True),
                                                           (F32, G14, (64, 96, 64, 40), False)):
        t = rs.randn(N, ci0 + ci1, D, H, W).astype(np.float32)                    # the shared bottleneck buffer: member k reads its slice
        ws = [(rs.randn(co, ci, 3, 3, 3) / np.sqrt(ci * 27)).astype(np.float32) for ci, co in ((ci0, co0), (ci1, co1))]
        aff = [((1 + 0.1 * rs.randn(co)).astype(np.float32), (0.2 * rs.randn(co)).astype(np.float32)) for co in (co0, co1)]
        te = bk.dev(encode(cl(t), dt))
        ctot = 8 + co0 + co1 + 8
        outs = []
        for grouped in (True, False):
            yb = bk.dev(np.zeros((N, D, H, W, ctot), NP_DT[dt]))
            keep, items, descs = [], (_capi.ConvItem * 2)(), []
            for k_, (ci, co, xoff, yoff) in enumerate(((ci0, co0, 0, 8), (ci1, co1, ci0, 8 + co0))):
                d = _capi.ConvDesc(dtype=dt, N=N, D=D, H=H, W=W, Cin=ci, Cout=co, kd=3, kh=3, kw=3, x_cstride=ci0 + ci1, x_coff=xoff,
                                   y_cstride=ctot, y_coff=yoff, res_cstride=0, res_coff=0, relu=1, split=0, y2_cstride=0, y2_coff=0)
                wp = pack_weight(bk, ws[k_], dt)
                sc, sh = bk.dev(aff[k_][0]), bk.dev(aff[k_][1])
                keep += [d, wp, sc, sh]
                descs.append(d)
                it = items[k_]
                it.desc = ctypes.pointer(d)
                it.x, it.w_packed, it.scale, it.shift, it.res, it.y = (ctypes.cast(te.ptr, ctypes.c_void_p).value, ctypes.cast(wp.ptr, ctypes.c_void_p).value,
                                                                       ctypes.cast(sc.ptr, ctypes.c_void_p).value, ctypes.cast(sh.ptr, ctypes.c_void_p).value, None,
                                                                       ctypes.cast(yb.ptr, ctypes.c_void_p).value)
            if grouped:
                assert bk.lib.step_conv_group_kernel_name(items, 2, buf, 256) == 0
                assert bool(buf.value) == merged, (dt, buf.value)
                if merged:
                    assert b"conv_tap_group_kernel" in buf.value
                assert bk.lib.step_conv_forward_group(items, 2, bk.stream) == 0
            else:
                for k_ in range(2):
                    it = items[k_]
                    assert bk.lib.step_conv_forward(it.desc, it.x, it.w_packed, it.scale, it.shift, None, it.y, None, bk.stream) == 0
            outs.append(yb.get())
        assert np.array_equal(outs[0], outs[1]), (dt, ci0, co0, ci1, co1)
        y = decode(outs[0], dt)
        assert not y[..., :8].any() and not y[..., 8 + co0 + co1:].any()
        for k_, (lo, hi, xlo, xhi) in enumerate(((8, 8 + co0, 0, ci0), (8 + co0, 8 + co0 + co1, ci0, ci0 + ci1))):
            ref = ref_conv(t[:, xlo:xhi], ws[k_], aff[k_][0], aff[k_][1], dt)
            got = uncl(y[..., lo:hi])
            assert np.abs(got - ref).max() / np.abs(ref).max() < tol(dt), (dt, k_)
    assert bk.lib.step_conv_forward_group(None, 0, bk.stream) == 0 and bk.lib.step_conv_forward_group(None, 1, bk.stream) == -3


def case_conv_group_with_pointwise_member(bk, golden):
    """step_conv_forward_group with a third, pointwise item (an Inception block's branch_3 conv on the pooled tensor): when the two
    3x3x3 members are fewer workgroups than the chip has CUs its 256-thread workgroups ride in the same grid
    (conv_tap_group_pw_kernel) -- bit-identical to three step_conv_forward calls, ragged pixel and channel counts included; with
    option conv_group_pw = 0, with a residual or with a member the planner sends elsewhere it is launched behind them."""
    rs = np.random.RandomState(53)
    buf = ctypes.create_string_buffer(256)
    N, D, H, W = 3, 8, 14, 14                                  # 4704 pixels: enough rows for the planner to stream the pointwise layers (conv_pw_kernel<T, 1, 4>)
    for dt, (ci0, co0, ci1, co1), (cp, cop), order, rides in ((BF16, (64, 72, 64, 40), (128, 96), (0, 1, 2), True), (F16, (64, 64, 64, 64), (136, 72), (2, 0, 1), True),
                                                              (BF16, (64, 40, 64, 72), (64, 128), (0, 2, 1), False)):    # (K = 64: the planner keeps the tiled kernel -- not carried)
        t = rs.randn(N, ci0 + ci1, D, H, W).astype(np.float32)
        pl = rs.randn(N, cp, D, H, W).astype(np.float32)                           # the pooled tensor: another buffer
        shapes = ((ci0, co0, 3), (ci1, co1, 3), (cp, cop, 1))
        ws = [(rs.randn(co, ci, k, k, k) / np.sqrt(ci * k ** 3)).astype(np.float32) for ci, co, k in shapes]
        aff = [((1 + 0.1 * rs.randn(co)).astype(np.float32), (0.2 * rs.randn(co)).astype(np.float32)) for _, co, _ in shapes]
        te, pe = bk.dev(encode(cl(t), dt)), bk.dev(encode(cl(pl), dt))
        ctot = 8 + co0 + co1 + cop + 8
        outs = {}
        for mode in ("group", "group_nopw", "group_throughput", "separate"):
            yb = bk.dev(np.zeros((N, D, H, W, ctot), NP_DT[dt]))
            keep, items = [], (_capi.ConvItem * 3)()
            specs = ((ci0, co0, 3, te, ci0 + ci1, 0, 8), (ci1, co1, 3, te, ci0 + ci1, ci0, 8 + co0), (cp, cop, 1, pe, cp, 0, 8 + co0 + co1))
            for slot, k_ in enumerate(order):
                ci, co, kk, xb, xcs, xoff, yoff = specs[k_]
                d = _capi.ConvDesc(dtype=dt, N=N, D=D, H=H, W=W, Cin=ci, Cout=co, kd=kk, kh=kk, kw=kk, x_cstride=xcs, x_coff=xoff,
                                   y_cstride=ctot, y_coff=yoff, res_cstride=0, res_coff=0, relu=1, split=0, y2_cstride=0, y2_coff=0)
                wp = pack_weight(bk, ws[k_], dt)
                sc, sh = bk.dev(aff[k_][0]), bk.dev(aff[k_][1])
                keep += [d, wp, sc, sh]
                it = items[slot]
                it.desc = ctypes.pointer(d)
                it.x, it.w_packed, it.scale, it.shift, it.res, it.y = (ctypes.cast(xb.ptr, ctypes.c_void_p).value, ctypes.cast(wp.ptr, ctypes.c_void_p).value,
                                                                       ctypes.cast(sc.ptr, ctypes.c_void_p).value, ctypes.cast(sh.ptr, ctypes.c_void_p).value, None,
                                                                       ctypes.cast(yb.ptr, ctypes.c_void_p).value)
            if mode == "separate":
                for slot in range(3):
                    it = items[slot]
                    assert bk.lib.step_conv_forward(it.desc, it.x, it.w_packed, it.scale, it.shift, None, it.y, None, bk.stream) == 0
            else:
                # (the `throughput` profile -- several batches in flight -- launches the pointwise member on its own as conv_group_pw = 0 does)
                with _capi.options(bk.lib, **(dict(throughput=1) if mode == "group_throughput" else dict(conv_group_pw=256 if mode == "group" else 0))):
                    assert bk.lib.step_conv_group_kernel_name(items, 3, buf, 256) == 0
                    assert (b"conv_tap_group_pw_kernel" in buf.value) == (mode == "group" and rides), (mode, buf.value)
                    assert b"conv_tap_group" in buf.value
                    assert bk.lib.step_conv_forward_group(items, 3, bk.stream) == 0
            outs[mode] = yb.get()
        assert np.array_equal(outs["group"], outs["separate"]) and np.array_equal(outs["group_nopw"], outs["separate"]), (dt, cp, cop)
        assert np.array_equal(outs["group_throughput"], outs["separate"]), (dt, cp, cop)
        y = decode(outs["group"], dt)
        assert not y[..., :8].any() and not y[..., ctot - 8:].any()
        ref = ref_conv(pl, ws[2], aff[2][0], aff[2][1], dt)
        got = uncl(y[..., 8 + co0 + co1:8 + co0 + co1 + cop])
        assert np.abs(got - ref).max() / np.abs(ref).max() < tol(dt), dt
    # two pointwise items, or a pointwise item beside ONE 3x3x3 conv: nothing to merge into -- separate launches, empty name
    assert bk.lib.step_conv_group_kernel_name(items, 2, buf, 256) == 0


def case_conv_forward_pre_matches_two_launches(bk, golden):
    """step_conv_forward_pre (conv3d_2b evaluated inside conv3d_2c's halo staging) against the two layers launched one after the
    other: bit-identical outputs on the 4 x 8 x 8 tile form (with its NB = 1 tail launch) and on general boxes, image borders and
    ragged tiles included; shapes outside the fused form's contract are refused with STEP_E_UNSUPPORTED."""
    rs = np.random.RandomState(57)
    info = (ctypes.c_int * 10)()
    seen = set()
    for dt, (N, D, H, W), Cout, opts in ((BF16, (1, 4, 24, 24), 192, dict(conv_slots=4, conv_nb=3, conv_waves=8)),
                                         (F16, (3, 8, 14, 14), 200, {}), (BF16, (1, 5, 21, 19), 96, dict(conv_waves=8))):
        x = rs.randn(N, 64, D, H, W).astype(np.float32)
        wa = (rs.randn(64, 64, 1, 1, 1) / 8).astype(np.float32)
        wb = (rs.randn(Cout, 64, 3, 3, 3) / np.sqrt(64 * 27)).astype(np.float32)
        sa, ha = (1 + 0.1 * rs.randn(64)).astype(np.float32), (0.2 * rs.randn(64)).astype(np.float32)
        sb, hb = (1 + 0.1 * rs.randn(Cout)).astype(np.float32), (0.2 * rs.randn(Cout)).astype(np.float32)
        xe = bk.dev(encode(cl(x), dt))
        wpa, wpb = pack_weight(bk, wa, dt), pack_weight(bk, wb, dt)
        dsa, dha, dsb, dhb = bk.dev(sa), bk.dev(ha), bk.dev(sb), bk.dev(hb)
        da = _capi.ConvDesc(dtype=dt, N=N, D=D, H=H, W=W, Cin=64, Cout=64, kd=1, kh=1, kw=1, x_cstride=64, x_coff=0, y_cstride=64, y_coff=0,
                            res_cstride=0, res_coff=0, relu=1, split=0, y2_cstride=0, y2_coff=0)
        db = _capi.ConvDesc(dtype=dt, N=N, D=D, H=H, W=W, Cin=64, Cout=Cout, kd=3, kh=3, kw=3, x_cstride=64, x_coff=0, y_cstride=Cout, y_coff=0,
                            res_cstride=0, res_coff=0, relu=1, split=0, y2_cstride=0, y2_coff=0)
        with _capi.options(bk.lib, **opts):
            assert bk.lib.step_conv_plan_info(ctypes.byref(db), info, 10) == 0
            mid = bk.dev(np.zeros((N, D, H, W, 64), NP_DT[dt]))
            y2 = bk.dev(np.zeros((N, D, H, W, Cout), NP_DT[dt]))
            assert bk.lib.step_conv_forward(ctypes.byref(da), xe.ptr, wpa.ptr, dsa.ptr, dha.ptr, None, mid.ptr, None, bk.stream) == 0
            assert bk.lib.step_conv_forward(ctypes.byref(db), mid.ptr, wpb.ptr, dsb.ptr, dhb.ptr, None, y2.ptr, None, bk.stream) == 0
            y1 = bk.dev(np.full((N, D, H, W, Cout), 7, NP_DT[dt]))
            rc = bk.lib.step_conv_forward_pre(ctypes.byref(db), xe.ptr, wpb.ptr, dsb.ptr, dhb.ptr, wpa.ptr, dsa.ptr, dha.ptr, 64, y1.ptr, bk.stream)
        fusable = info[0] == 1 and info[4] == 1 and info[3] == 8 and info[1] in (0, 3)
        assert rc == (0 if fusable else -4), (rc, list(info))
        if fusable:
            seen.add(info[1])
            assert np.array_equal(y1.get(), y2.get()), (dt, N, D, H, W, Cout, list(info))
            ref = ref_conv(np.maximum(ref_conv(x, wa, sa, ha, dt), 0).astype(np.float32), wb, sb, hb, dt)
            got = uncl(decode(y1.get(), dt))
            assert np.abs(got - ref).max() / np.abs(ref).max() < 3 * tol(dt), dt
    assert seen == {0, 3}, seen                                  # both tile forms were exercised
    # outside the contract: a 32-channel pointwise layer, fp32 storage, a null pointwise weight
    assert bk.lib.step_conv_forward_pre(ctypes.byref(db), xe.ptr, wpb.ptr, dsb.ptr, dhb.ptr, wpa.ptr, dsa.ptr, dha.ptr, 32, y1.ptr, bk.stream) == -4
    assert bk.lib.step_conv_forward_pre(ctypes.byref(db), xe.ptr, wpb.ptr, dsb.ptr, dhb.ptr, None, dsa.ptr, dha.ptr, 64, y1.ptr, bk.stream) == -3
    df = _capi.ConvDesc(dtype=F32, N=1, D=4, H=8, W=8, Cin=64, Cout=64, kd=3, kh=3, kw=3, x_cstride=64, x_coff=0, y_cstride=64, y_coff=0,
                        res_cstride=0, res_coff=0, relu=1, split=0, y2_cstride=0, y2_coff=0)
    assert bk.lib.step_conv_forward_pre(ctypes.byref(df), xe.ptr, wpb.ptr, dsb.ptr, dhb.ptr, wpa.ptr, dsa.ptr, dha.ptr, 64, y1.ptr, bk.stream) == -4


def case_conv_forward_pre_pool_matches_separate_calls(bk, golden):
    """step_conv_forward_pre_pool (conv3d_2b -> conv3d_2c -> maxPool3d_3a as one call: the (1,3,3) / (1,2,2) max pool taken on the conv's
    4-plane 8x8 tiles, seams completed by pool_seam_fix_kernel) against step_conv_forward_pre + step_maxpool3d_tf: BIT-IDENTICAL pooled
    tensors -- several tiles in both directions (row seams, column seams, corners), the NB = 1 launch of the partial last round, partial
    tiles and partial plane groups, odd map sides (the ceil-mode window that hangs over the zero pad), the pooled tensor as a channel
    slice of a wider buffer; layers the planner does not tile 4 x 8 x 8, no ReLU and fp32 are refused (workspace size 0, -4)."""
    rs = np.random.RandomState(61)
    info = (ctypes.c_int * 10)()
    ran = 0
    # (round 6) the PERSISTENT tile loop of the NB = 3 launch (conv_tap_pre_pool_persist_kernel: `conv_slots` workgroups walk the tiles):
    # 16 tiles on 8 workgroups + an NB = 1 tail launch of 2; 27 tiles in ONE launch of 8 workgroups (32 virtual ids: padding ids are skipped,
    # workgroups run 3 or 4 tiles); and the same layer one workgroup per tile (conv_persist = 0) -- all bit-identical to the separate calls
    for dt, (N, D, H, W), Cout, opts, ycs, yco in ((BF16, (1, 4, 24, 24), 192, dict(conv_slots=4, conv_nb=3, conv_waves=8), 192, 0),
                                                   (BF16, (1, 8, 24, 24), 192, dict(conv_slots=8, conv_nb=3, conv_waves=8, conv_gen=0), 192, 0),
                                                   (F16, (1, 12, 24, 20), 192, dict(conv_slots=8, conv_nb=3, conv_waves=8, conv_gen=0), 200, 8),
                                                   (BF16, (1, 8, 24, 24), 192, dict(conv_slots=8, conv_nb=3, conv_waves=8, conv_gen=0, conv_persist=0), 192, 0),
                                                   (F16, (1, 7, 21, 24), 72, dict(conv_gen=0, conv_waves=8), 88, 8),
                                                   (BF16, (1, 8, 16, 8), 64, dict(conv_gen=0, conv_waves=8), 64, 0)):
        x = rs.randn(N, 64, D, H, W).astype(np.float32)
        wa = (rs.randn(64, 64, 1, 1, 1) / 8).astype(np.float32)
        wb = (rs.randn(Cout, 64, 3, 3, 3) / np.sqrt(64 * 27)).astype(np.float32)
        sa, ha = (1 + 0.1 * rs.randn(64)).astype(np.float32), (0.2 * rs.randn(64)).astype(np.float32)
        sb, hb = (1 + 0.1 * rs.randn(Cout)).astype(np.float32), (0.2 * rs.randn(Cout)).astype(np.float32)
        xe = bk.dev(encode(cl(x), dt))
        wpa, wpb = pack_weight(bk, wa, dt), pack_weight(bk, wb, dt)
        dsa, dha, dsb, dhb = bk.dev(sa), bk.dev(ha), bk.dev(sb), bk.dev(hb)
        Hp, Wp = bk.lib.step_pool_out_size(H, 3, 2), bk.lib.step_pool_out_size(W, 3, 2)
        db = _capi.ConvDesc(dtype=dt, N=N, D=D, H=H, W=W, Cin=64, Cout=Cout, kd=3, kh=3, kw=3, x_cstride=64, x_coff=0, y_cstride=Cout, y_coff=0,
                            res_cstride=0, res_coff=0, relu=1, split=0, y2_cstride=0, y2_coff=0)
        dp = _capi.ConvDesc(dtype=dt, N=N, D=D, H=H, W=W, Cin=64, Cout=Cout, kd=3, kh=3, kw=3, x_cstride=64, x_coff=0, y_cstride=ycs, y_coff=yco,
                            res_cstride=0, res_coff=0, relu=1, split=0, y2_cstride=0, y2_coff=0)
        with _capi.options(bk.lib, **opts):
            assert bk.lib.step_conv_plan_info(ctypes.byref(db), info, 10) == 0
            assert info[0] == 1 and info[1] == 3 and info[4] == 1, list(info)         # conv_tap, the 4 x 8 x 8 tile, two-phase
            full = bk.dev(np.zeros((N, D, H, W, Cout), NP_DT[dt]))
            assert bk.lib.step_conv_forward_pre(ctypes.byref(db), xe.ptr, wpb.ptr, dsb.ptr, dhb.ptr, wpa.ptr, dsa.ptr, dha.ptr, 64, full.ptr, bk.stream) == 0
            want = bk.dev(np.full((N, D, Hp, Wp, ycs), 3, NP_DT[dt]))
            assert bk.lib.step_maxpool3d_tf(dt, full.ptr, N, D, H, W, Cout, Cout, 0, 1, 3, 3, 1, 2, 2, want.ptr, ycs, yco, bk.stream) == 0
            nb = bk.lib.step_conv_pre_pool_workspace_bytes(ctypes.byref(dp))
            assert nb > 0 and nb % 16 == 0
            ws = bk.dev(np.full(nb // 2, 0x7f7f, np.uint16))                      # garbage: every byte the fix kernel reads must have been written
            got = bk.dev(np.full((N, D, Hp, Wp, ycs), 3, NP_DT[dt]))
            assert bk.lib.step_conv_forward_pre_pool(ctypes.byref(dp), xe.ptr, wpb.ptr, dsb.ptr, dhb.ptr, wpa.ptr, dsa.ptr, dha.ptr, 64, got.ptr,
                                                     ws.ptr, nb, bk.stream) == 0
            assert bk.lib.step_conv_forward_pre_pool(ctypes.byref(dp), xe.ptr, wpb.ptr, dsb.ptr, dhb.ptr, wpa.ptr, dsa.ptr, dha.ptr, 64, got.ptr,
                                                     ws.ptr, nb - 16, bk.stream) == -2
            # ... and in its two parts (conv launches, then the seam pass): the same bits
            got2 = bk.dev(np.full((N, D, Hp, Wp, ycs), 3, NP_DT[dt]))
            ws2 = bk.dev(np.full(nb // 2, 0x7f7f, np.uint16))
            assert bk.lib.step_conv_forward_pre_pool_tiles(ctypes.byref(dp), xe.ptr, wpb.ptr, dsb.ptr, dhb.ptr, wpa.ptr, dsa.ptr, dha.ptr, 64, got2.ptr,
                                                           ws2.ptr, nb, bk.stream) == 0
            assert bk.lib.step_conv_pre_pool_finish(ctypes.byref(dp), got2.ptr, ws2.ptr, nb, bk.stream) == 0
        g_, w_ = got.get(), want.get()
        assert np.array_equal(got2.get(), g_)
        assert np.array_equal(g_, w_), (dt, N, D, H, W, Cout, int((g_ != w_).sum()), np.argwhere(g_ != w_)[:4].tolist())
        assert (decode(g_[..., yco:yco + Cout], dt) >= 0).all()
        ran += 1
    assert ran == 6
    # outside the contract: a general-box layer (the 14 x 14 map), no ReLU, fp32
    d14 = _capi.ConvDesc(dtype=BF16, N=3, D=8, H=14, W=14, Cin=64, Cout=64, kd=3, kh=3, kw=3, x_cstride=64, x_coff=0, y_cstride=64, y_coff=0,
                         res_cstride=0, res_coff=0, relu=1, split=0, y2_cstride=0, y2_coff=0)
    assert bk.lib.step_conv_plan_info(ctypes.byref(d14), info, 10) == 0
    if info[1] != 3:
        assert bk.lib.step_conv_pre_pool_workspace_bytes(ctypes.byref(d14)) == 0
    dn = _capi.ConvDesc(dtype=BF16, N=1, D=4, H=24, W=24, Cin=64, Cout=192, kd=3, kh=3, kw=3, x_cstride=64, x_coff=0, y_cstride=192, y_coff=0,
                        res_cstride=0, res_coff=0, relu=0, split=0, y2_cstride=0, y2_coff=0)
    df = _capi.ConvDesc(dtype=F32, N=1, D=4, H=24, W=24, Cin=64, Cout=192, kd=3, kh=3, kw=3, x_cstride=64, x_coff=0, y_cstride=192, y_coff=0,
                        res_cstride=0, res_coff=0, relu=1, split=0, y2_cstride=0, y2_coff=0)
    for dbad in (dn, df):
        assert bk.lib.step_conv_pre_pool_workspace_bytes(ctypes.byref(dbad)) == 0
        assert bk.lib.step_conv_forward_pre_pool(ctypes.byref(dbad), xe.ptr, wpb.ptr, dsb.ptr, dhb.ptr, wpa.ptr, dsa.ptr, dha.ptr, 64, got.ptr,
                                                 ws.ptr, nb, bk.stream) == -4


def case_pool_conv_forward_matches_two_launches(bk, golden):
    """step_pool_conv_forward (an Inception block's 3x3x3 / 1 max pool + its fused pointwise triple in ONE grid) against
    step_maxpool3d_tf + step_conv_forward: bit-identical pooled tensor and conv outputs (two destinations through `split`, ragged
    pixel and channel counts); layers the planner does not stream at NB = 1 and fp32 are refused with STEP_E_UNSUPPORTED."""
    rs = np.random.RandomState(59)
    info = (ctypes.c_int * 10)()
    ran = 0
    for dt, (N, D, H, W), Cin, Cout, split in ((BF16, (8, 3, 14, 14), 136, 104, 40), (F16, (8, 3, 14, 14), 128, 96, 0), (BF16, (1, 3, 14, 14), 128, 96, 0)):
        x = rs.randn(N, Cin, D, H, W).astype(np.float32)
        w = (rs.randn(Cout, Cin, 1, 1, 1) / np.sqrt(Cin)).astype(np.float32)
        sc, sh = (1 + 0.1 * rs.randn(Cout)).astype(np.float32), (0.2 * rs.randn(Cout)).astype(np.float32)
        xe = bk.dev(encode(cl(x), dt))
        wp, dsc, dsh = pack_weight(bk, w, dt), bk.dev(sc), bk.dev(sh)
        c0 = split if split else Cout
        d = _capi.ConvDesc(dtype=dt, N=N, D=D, H=H, W=W, Cin=Cin, Cout=Cout, kd=1, kh=1, kw=1, x_cstride=Cin, x_coff=0, y_cstride=c0 + 8, y_coff=8,
                           res_cstride=0, res_coff=0, relu=1, split=split, y2_cstride=Cout - split if split else 0, y2_coff=0)
        assert bk.lib.step_conv_plan_info(ctypes.byref(d), info, 10) == 0
        outs = []
        for fused in (True, 2, 1, False):                    # True: the library's choice; 2 | 1: 128- | 64-channel pointwise workgroups forced (option conv_nb)
            py = bk.dev(np.full((N, D, H, W, Cin), 3, NP_DT[dt]))
            y = bk.dev(np.zeros((N, D, H, W, c0 + 8), NP_DT[dt]))
            y2 = bk.dev(np.zeros((N, D, H, W, max(Cout - split, 1)), NP_DT[dt]))
            if fused:
                _capi.set_option(bk.lib, "conv_nb", 0 if fused is True else fused)
                want_nb = bk.lib.step_pool_conv_plan_nb(ctypes.byref(d))
                rc = bk.lib.step_pool_conv_forward(dt, xe.ptr, N, D, H, W, Cin, Cin, 0, py.ptr, Cin, 0, ctypes.byref(d), xe.ptr, wp.ptr, dsc.ptr, dsh.ptr,
                                                   y.ptr, y2.ptr if split else None, bk.stream)
                _capi.set_option(bk.lib, "conv_nb", 0)
                ok = info[0] == 2 and info[2] <= 2
                assert rc == (0 if ok else -4), (rc, list(info))
                assert want_nb == ((1 if fused is True else fused) if ok else 0), (want_nb, fused, list(info))     # (these maps are small: 64-channel workgroups unless forced)
                if not ok:
                    break
            else:
                assert bk.lib.step_maxpool3d_tf(dt, xe.ptr, N, D, H, W, Cin, Cin, 0, 3, 3, 3, 1, 1, 1, py.ptr, Cin, 0, bk.stream) == 0
                assert bk.lib.step_conv_forward(ctypes.byref(d), xe.ptr, wp.ptr, dsc.ptr, dsh.ptr, None, y.ptr, y2.ptr if split else None, bk.stream) == 0
            outs.append((py.get(), y.get(), y2.get()))
        if len(outs) == 4:
            ran += 1
            for o in outs[:3]:
                for a, b in zip(o, outs[3]):
                    assert np.array_equal(a, b), (dt, N, Cin, Cout, split)
            assert not decode(outs[0][1], dt)[..., :8].any()
    assert ran >= 2, ran
    df = _capi.ConvDesc(dtype=F32, N=1, D=3, H=14, W=14, Cin=128, Cout=96, kd=1, kh=1, kw=1, x_cstride=128, x_coff=0, y_cstride=96, y_coff=0,
                        res_cstride=0, res_coff=0, relu=1, split=0, y2_cstride=0, y2_coff=0)
    assert bk.lib.step_pool_conv_forward(F32, xe.ptr, 1, 3, 14, 14, 128, 128, 0, py.ptr, 128, 0, ctypes.byref(df), xe.ptr, wp.ptr, dsc.ptr, dsh.ptr,
                                         y.ptr, None, bk.stream) == -4


def case_mfma_clock_probe(bk, golden):
    """step_mfma_clock_probe (the diagnostic behind bench.py's `sustained_on_this_box`): every workgroup reports its loop; on the
    interpreter the counters read 0, on the GPU the clock lies in the part's range and the matrix pipe issues one 32x32x16 per
    32 cycles per SIMD."""
    wgs, iters = 8, 300
    out = bk.dev(np.zeros(3 * wgs, np.uint64))
    assert bk.lib.step_mfma_clock_probe(out.ptr, wgs, iters, bk.stream) == 0
    h = out.get().reshape(wgs, 3)
    assert (h[:, 2] == 1).all()
    if bk.name == "gfx950":
        ghz = h[:, 0] / (h[:, 1] * 10.0)
        assert (ghz > 0.5).all() and (ghz < 3.0).all(), ghz
        assert (h[:, 0] >= 4 * iters * 32 * 0.9).all(), h[:, 0]            # 4 MFMAs x 32 cycles per iteration at least
    else:
        assert not h[:, :2].any()
    assert bk.lib.step_mfma_clock_probe(None, 0, 10, bk.stream) == 0 and bk.lib.step_mfma_clock_probe(None, 1, 10, bk.stream) == -3
    assert bk.lib.step_mfma_clock_probe(out.ptr, -1, 10, bk.stream) < 0


def case_hbm_stream_probe(bk, golden):
    """step_hbm_stream_probe (bench.py's measured HBM ceiling): the copy is exact for any workgroup count, incl. fewer lanes than
    vectors and a ragged last pass; argument checks."""
    rs = np.random.RandomState(5)
    src = rs.randint(0, 1 << 30, size=4 * 1237).astype(np.int32)          # 1237 16-byte vectors
    a = bk.dev(src)
    for wgs in (1, 3, 64):
        b = bk.dev(np.zeros_like(src))
        assert bk.lib.step_hbm_stream_probe(a.ptr, b.ptr, src.nbytes, wgs, bk.stream) == 0
        assert np.array_equal(b.get(), src)
    b = bk.dev(np.zeros_like(src))
    assert bk.lib.step_hbm_stream_probe(a.ptr, b.ptr, 0, 4, bk.stream) == 0 and not b.get().any()
    assert bk.lib.step_hbm_stream_probe(a.ptr, b.ptr, 24, 4, bk.stream) == -2
    assert bk.lib.step_hbm_stream_probe(None, b.ptr, 32, 4, bk.stream) == -3
    assert bk.lib.step_hbm_stream_probe(getattr(a.ptr, "value", a.ptr) + 4, b.ptr, 32, 4, bk.stream) == -5


def case_conv_tail_round_split(bk, golden):
    """A layer that is one channel group deep and whose pixel tiles end in a small partial round of one-workgroup-per-CU slots is
    launched in two parts: the full rounds at NB = 3 and the tail tiles at NB = 1 (three times as many, shorter workgroups).
    At interpreter size with option conv_slots = 4 (9 tiles: two rounds + one tile): same result, bit for bit, as the single launch."""
    rs = np.random.RandomState(43)
    N, Cin, Cout, D, H, W = 1, 64, 192, 4, 24, 24
    x = rs.randn(N, Cin, D, H, W).astype(np.float32)
    w = (rs.randn(Cout, Cin, 3, 3, 3) / np.sqrt(Cin * 27)).astype(np.float32)
    scale = (1 + 0.1 * rs.randn(Cout)).astype(np.float32)
    shift = (0.2 * rs.randn(Cout)).astype(np.float32)
    try:
        for k_, v_ in (("conv_slots", 4), ("conv_nb", 3), ("conv_waves", 8)):
            _capi.set_option(bk.lib, k_, v_)
        for dt in (BF16, F32):
            d = _capi.ConvDesc(dtype=dt, N=N, D=D, H=H, W=W, Cin=Cin, Cout=Cout, kd=3, kh=3, kw=3, x_cstride=Cin, x_coff=0, y_cstride=Cout, y_coff=0,
                               res_cstride=0, res_coff=0, relu=1, split=0, y2_cstride=0, y2_coff=0)
            info = (ctypes.c_int * 10)()
            assert bk.lib.step_conv_plan_info(ctypes.byref(d), info, 10) == 0
            assert info[0] == 1 and info[2] == 3 and info[3] == 8 and info[9] == 9, list(info)     # 9 tiles, one channel group
            ref = ref_conv(x, w, scale, shift, dt)
            outs = {}
            for mode in ("1", "0"):
                _capi.set_option(bk.lib, "conv_tail", int(mode))
                outs[mode] = run_conv(bk, x, w, scale, shift, dt)
                assert np.abs(outs[mode] - ref).max() / np.abs(ref).max() < tol(dt), (mode, dt)
            assert np.array_equal(outs["1"], outs["0"]), dt
    finally:
        for k_, v_ in (("conv_slots", 0), ("conv_nb", 0), ("conv_waves", 0), ("conv_tail", 1)):
            _capi.set_option(bk.lib, k_, v_)


def case_conv_pointwise_weight_stationary(bk, golden):
    """conv_pws_kernel (the weight-stationary short-K pointwise stream, option conv_pws = 1) against the oracle and against the
    default kernel: one to four 64-channel steps, a K tail that is not a multiple of 64 or 32, one to four passes over the
    channel blocks with a partial last block, ragged pixel count, input and output channel slices, two destinations, no ReLU / no
    affine, a residual tensor (a channel slice of a wider one; round 6: the heads' Bottleneck conv3); K > 256 stays with the default
    kernels."""
    rs = np.random.RandomState(31)
    # (N, Cin, Cout, D, H, W, split, relu, affine, x_pad, y_pad[, res_pad])
    cases = ((1, 64, 64, 2, 16, 33, 0, True, True, (0, 0), (0, 0)),
             (1, 192, 176, 1, 20, 53, 64, True, True, (8, 16), (8, 8)),
             (1, 528, 128, 1, 32, 33, 0, True, True, (0, 0), (0, 0)),
             (1, 144, 40, 2, 8, 67, 0, False, False, (0, 8), (16, 0)),
             (2, 32, 200, 1, 24, 23, 0, True, True, (0, 0), (0, 0)),
             (1, 64, 296, 1, 16, 65, 96, True, True, (0, 0), (0, 0)),          # 10 blocks: four passes
             (1, 256, 328, 1, 16, 65, 0, True, True, (0, 0), (0, 0)),          # 11 blocks x 16 chunks > 152: two workgroup-level channel groups
             (3, 256, 328, 1, 7, 49, 0, True, False, (0, 0), (0, 0), (8, 24)),  # residual: the heads' conv3 (no affine), 7-row maps
             (1, 96, 72, 2, 9, 61, 0, False, True, (8, 0), (0, 8), (0, 0)),     # residual, no ReLU, partial last block
             (1, 192, 176, 1, 20, 53, 64, True, True, (0, 0), (8, 8), (16, 0)))  # residual and two destinations
    try:
        for case in cases:
            (N, Cin, Cout, D, H, W, split, relu, affine, xp, yp), rp = case[:11], (case[11] if len(case) > 11 else None)
            x = rs.randn(N, Cin, D, H, W).astype(np.float32)
            w = (rs.randn(Cout, Cin, 1, 1, 1) / np.sqrt(Cin)).astype(np.float32)
            scale = (1 + 0.1 * rs.randn(Cout)).astype(np.float32) if affine else None
            shift = (0.2 * rs.randn(Cout)).astype(np.float32) if affine else None
            for dt in (BF16, F16):
                r = rs.randn(N, Cout, D, H, W).astype(np.float32) if rp is not None else None
                ref = ref_conv(x, w, scale, shift, dt, relu=relu, res=r)
                re_ = None
                if rp is not None:
                    rb = np.full((N, D, H, W, rp[0] + Cout + rp[1]), -77.0, np.float32)
                    rb[..., rp[0]:rp[0] + Cout] = cl(r)
                    re_ = bk.dev(encode(rb, dt))
                xb = np.full((N, D, H, W, xp[0] + Cin + xp[1]), 33.0, np.float32)
                xb[..., xp[0]:xp[0] + Cin] = cl(x)
                xe = bk.dev(encode(xb, dt))
                wp = pack_weight(bk, w, dt)
                ca = split if split else Cout
                sc, sh = bk.dev(scale), bk.dev(shift)
                outs = {}
                for mode in ("1", "0", "16"):                                   # "16": the stream with sixteen waves per workgroup (one operand set)
                    _capi.set_option(bk.lib, "conv_pws", int(mode != "0"))
                    _capi.set_option(bk.lib, "conv_pws_waves", 16 if mode == "16" else 8)
                    ya = bk.dev(np.zeros((N, D, H, W, yp[0] + ca + yp[1]), NP_DT[dt]))
                    yb = bk.dev(np.zeros((N, D, H, W, 8 + (Cout - ca)), NP_DT[dt]))
                    d = _capi.ConvDesc(dtype=dt, N=N, D=D, H=H, W=W, Cin=Cin, Cout=Cout, kd=1, kh=1, kw=1, x_cstride=xb.shape[-1], x_coff=xp[0],
                                       y_cstride=yp[0] + ca + yp[1], y_coff=yp[0], res_cstride=(rp[0] + Cout + rp[1]) if rp is not None else 0,
                                       res_coff=rp[0] if rp is not None else 0, relu=int(relu), split=split,
                                       y2_cstride=8 + (Cout - ca), y2_coff=8)
                    name = ctypes.create_string_buffer(256)
                    assert bk.lib.step_conv_kernel_name(ctypes.byref(d), name, 256) == 0
                    assert (b"conv_pws_kernel" in name.value) == (mode != "0" and Cin <= 256), (mode, name.value)   # (K <= 256: the whole K of a pixel group lives in registers)
                    if mode != "0" and Cin <= 256:
                        assert (b", 8, true>(" if rp is not None else (b", 16, false>(" if mode == "16" else b", 8, false>(")) in name.value, name.value   # (the residual form: eight waves)
                    assert bk.lib.step_conv_forward(ctypes.byref(d), xe.ptr, wp.ptr, sc.ptr, sh.ptr, re_.ptr if re_ is not None else None, ya.ptr,
                                                    yb.ptr if split else None, bk.stream) == 0
                    a = decode(ya.get(), dt)
                    assert not a[..., :yp[0]].any() and not a[..., yp[0] + ca:].any()
                    got = uncl(a[..., yp[0]:yp[0] + ca])
                    if split:
                        b = decode(yb.get(), dt)
                        assert not b[..., :8].any()
                        got = np.concatenate([got, uncl(b[..., 8:])], 1)
                    outs[mode] = got
                    assert np.abs(got - ref).max() / np.abs(ref).max() < tol(dt), (mode, Cin, Cout, dt)
                # same operands, same fp32 accumulation order along K: the two kernels agree to the last bit
                assert np.array_equal(outs["1"], outs["0"]) and np.array_equal(outs["1"], outs["16"]), (Cin, Cout, dt)
    finally:
        _capi.set_option(bk.lib, "conv_pws", -1)
        _capi.set_option(bk.lib, "conv_pws_waves", 0)


def case_conv_pointwise_cat(bk, golden):
    """step_conv_forward_cat (conv_pw2_kernel): a pointwise conv over the channel concat of two tensors that is never materialised -- against
    the oracle's conv over the concatenated input, and bit-identical to step_conv_forward on a materialised concat (same K order); whole
    and ragged tiles, a K tail, channel slices of wider tensors for both sources, residual / no affine, 4 and 8 waves, NB 1..3; the shapes
    the form does not cover are refused (STEP_E_UNSUPPORTED) so that the caller launches the halves."""
    rs = np.random.RandomState(77)
    # (N, Ca, Cb, Cout, D, H, W, relu, affine, with_res, xa_pad, xb_pad)
    cases = ((1, 96, 72, 136, 1, 64, 65, True, True, True, (32, 8), (8, 16)),      # ragged pixel tile, K = 168 (tail of 8 channels), residual
             (1, 64, 64, 128, 1, 64, 64, False, False, False, (0, 0), (0, 0)))      # whole tiles, whole K steps: the mask-free loop
    try:
        for (N, Ca, Cb, Cout, D, H, W, relu, affine, with_res, ap, bp) in cases:
            xa = rs.randn(N, Ca, D, H, W).astype(np.float32)
            xb = rs.randn(N, Cb, D, H, W).astype(np.float32)
            w = (rs.randn(Cout, Ca + Cb, 1, 1, 1) / np.sqrt(Ca + Cb)).astype(np.float32)
            scale = (1 + 0.1 * rs.randn(Cout)).astype(np.float32) if affine else None
            shift = (0.2 * rs.randn(Cout)).astype(np.float32) if affine else None
            r = rs.randn(N, Cout, D, H, W).astype(np.float32) if with_res else None
            xc = np.concatenate([xa, xb], 1)
            for dt in (BF16, F16):
                ref = ref_conv(xc, w, scale, shift, dt, relu=relu, res=r)
                ab = np.full((N, D, H, W, ap[0] + Ca + ap[1]), 33.0, np.float32)
                ab[..., ap[0]:ap[0] + Ca] = cl(xa)
                bb = np.full((N, D, H, W, bp[0] + Cb + bp[1]), -44.0, np.float32)
                bb[..., bp[0]:bp[0] + Cb] = cl(xb)
                ae, be, ce = bk.dev(encode(ab, dt)), bk.dev(encode(bb, dt)), bk.dev(encode(cl(xc), dt))
                re_ = bk.dev(encode(cl(r), dt)) if with_res else None
                wp = pack_weight(bk, w, dt)
                sc, sh = bk.dev(scale), bk.dev(shift)

                def desc(cstride, coff):
                    return _capi.ConvDesc(dtype=dt, N=N, D=D, H=H, W=W, Cin=Ca + Cb, Cout=Cout, kd=1, kh=1, kw=1, x_cstride=cstride, x_coff=coff,
                                          y_cstride=Cout, y_coff=0, res_cstride=Cout if with_res else 0, res_coff=0, relu=int(relu), split=0,
                                          y2_cstride=0, y2_coff=0)
                for waves, nb in ((8, 0), (8, 3), (8, 1), (4, 0), (4, 3)):
                    _capi.set_option(bk.lib, "conv_waves", waves)
                    _capi.set_option(bk.lib, "conv_nb", nb)
                    y1 = bk.dev(np.zeros((N, D, H, W, Cout), NP_DT[dt]))
                    d = desc(ab.shape[-1], ap[0])
                    rc = bk.lib.step_conv_forward_cat(ctypes.byref(d), ae.ptr, Ca, be.ptr, bb.shape[-1], bp[0], wp.ptr, sc.ptr, sh.ptr,
                                                      re_.ptr if re_ is not None else None, y1.ptr, None, bk.stream)
                    assert rc == 0, (rc, waves, nb)
                    got = uncl(decode(y1.get(), dt))
                    assert np.abs(got - ref).max() / np.abs(ref).max() < tol(dt), (Ca, Cb, dt, waves, nb)
                    y2 = bk.dev(np.zeros((N, D, H, W, Cout), NP_DT[dt]))
                    dc = desc(Ca + Cb, 0)
                    assert bk.lib.step_conv_forward(ctypes.byref(dc), ce.ptr, wp.ptr, sc.ptr, sh.ptr, re_.ptr if re_ is not None else None, y2.ptr,
                                                    None, bk.stream) == 0
                    assert np.array_equal(got, uncl(decode(y2.get(), dt))), (Ca, Cb, dt, waves, nb)      # one accumulation over the same K order
                _capi.set_option(bk.lib, "conv_waves", 0)
                _capi.set_option(bk.lib, "conv_nb", 0)
                # refused: a first source that does not end on a 32-channel K step; too few pixels for the streaming GEMM
                y1 = bk.dev(np.zeros((N, D, H, W, Cout), NP_DT[dt]))
                d = desc(ab.shape[-1], ap[0])
                d.res_cstride = 0
                rc = bk.lib.step_conv_forward_cat(ctypes.byref(d), ae.ptr, Ca - 8, be.ptr, bb.shape[-1], 0, wp.ptr, sc.ptr, sh.ptr, None, y1.ptr,
                                                  None, bk.stream)
                assert rc == (-4 if bp[0] + bp[1] >= 8 else -2), rc          # (-2: the second source has no room for 8 more channels)
                ds = _capi.ConvDesc(dtype=dt, N=1, D=1, H=4, W=4, Cin=Ca + Cb, Cout=Cout, kd=1, kh=1, kw=1, x_cstride=ab.shape[-1], x_coff=ap[0],
                                    y_cstride=Cout, y_coff=0, res_cstride=0, res_coff=0, relu=1, split=0, y2_cstride=0, y2_coff=0)
                assert bk.lib.step_conv_forward_cat(ctypes.byref(ds), ae.ptr, Ca, be.ptr, bb.shape[-1], bp[0], wp.ptr, sc.ptr, sh.ptr, None, y1.ptr,
                                                    None, bk.stream) == -4
        # fp32 storage: no such form
        d32 = _capi.ConvDesc(dtype=F32, N=1, D=1, H=64, W=64, Cin=128, Cout=128, kd=1, kh=1, kw=1, x_cstride=64, x_coff=0, y_cstride=128, y_coff=0,
                             res_cstride=0, res_coff=0, relu=1, split=0, y2_cstride=0, y2_coff=0)
        z = bk.dev(np.zeros((1, 1, 64, 64, 128), np.float32))
        wz = bk.dev(np.zeros(bk.lib.step_conv_packed_elems(128, 128, 1, 1, 1), np.float32))
        assert bk.lib.step_conv_forward_cat(ctypes.byref(d32), z.ptr, 64, z.ptr, 128, 64, wz.ptr, None, None, None, z.ptr, None, bk.stream) == -4
    finally:
        _capi.set_option(bk.lib, "conv_waves", 0)
        _capi.set_option(bk.lib, "conv_nb", 0)


def case_conv_pointwise_eight_wave_ring(bk, golden):
    """conv_pw_kernel with eight waves (round 6: a four-deep slab ring in LDS, one barrier per TWO K steps) on every loop shape -- 1, 2, 3, 4, 5
    and 7 K steps (odd counts end on a single step), whole and ragged pixel tiles, a K tail -- against the oracle and BIT-IDENTICAL to the
    four-wave form (three-deep ring, one barrier per step): same slabs, same order of the multiplications."""
    rs = np.random.RandomState(83)
    try:
        for (Cin, Cout, H, W, relu) in ((32, 64, 64, 64, True), (64, 128, 64, 65, True), (96, 96, 64, 64, False), (128, 64, 32, 128, True),
                                        (160, 128, 64, 65, True), (200, 72, 64, 64, True)):
            x = rs.randn(1, Cin, 1, H, W).astype(np.float32)
            w = (rs.randn(Cout, Cin, 1, 1, 1) / np.sqrt(Cin)).astype(np.float32)
            scale, shift = (1 + 0.1 * rs.randn(Cout)).astype(np.float32), (0.2 * rs.randn(Cout)).astype(np.float32)
            for dt in (BF16, F16):
                ref = ref_conv(x, w, scale, shift, dt, relu=relu)
                xe, wp, sc, sh = bk.dev(encode(cl(x), dt)), pack_weight(bk, w, dt), bk.dev(scale), bk.dev(shift)
                d = _capi.ConvDesc(dtype=dt, N=1, D=1, H=H, W=W, Cin=Cin, Cout=Cout, kd=1, kh=1, kw=1, x_cstride=Cin, x_coff=0, y_cstride=Cout, y_coff=0,
                                   res_cstride=0, res_coff=0, relu=int(relu), split=0, y2_cstride=0, y2_coff=0)
                outs = {}
                for waves in (8, 4):
                    _capi.set_option(bk.lib, "conv_impl", 5)
                    _capi.set_option(bk.lib, "conv_waves", waves)
                    name = ctypes.create_string_buffer(256)
                    assert bk.lib.step_conv_kernel_name(ctypes.byref(d), name, 256) == 0
                    assert b"conv_pw_kernel" in name.value and (b", %d>(" % waves) in name.value, name.value
                    y = bk.dev(np.zeros((1, 1, H, W, Cout), NP_DT[dt]))
                    assert bk.lib.step_conv_forward(ctypes.byref(d), xe.ptr, wp.ptr, sc.ptr, sh.ptr, None, y.ptr, None, bk.stream) == 0
                    outs[waves] = uncl(decode(y.get(), dt))
                    assert np.abs(outs[waves] - ref).max() / np.abs(ref).max() < tol(dt), (Cin, Cout, dt, waves)
                assert np.array_equal(outs[8], outs[4]), (Cin, Cout, dt)
    finally:
        _capi.set_option(bk.lib, "conv_impl", -1)
        _capi.set_option(bk.lib, "conv_waves", 0)


def case_detect_compact(bk, golden):
    """step_detect_compact (the rows test.py:196-204 appends behind the NMS, all iterations and clips in one launch) against a numpy
    restatement: per (iteration, clip) the set flags of keep[i, b] in row-major order -- classes ascending, tubes ascending -- with
    box / [W,H,W,H] (IEEE division, bit-exact), the tube's score of that class, class and tube index; ragged clips, an empty clip,
    an all-zero and an all-one mask, flag counts that are not a multiple of the workgroup; no flags at all."""
    rs = np.random.RandomState(101)
    for (I, nums, NC, kmax, dens) in ((3, (11, 0, 7, 34), 60, 34, 0.05), (1, (5,), 3, 7, 1.0), (2, (64, 3), 17, 64, 0.0), (8, (2, 2), 300, 2, 0.5)):
        B, N = len(nums), max(sum(nums), 1)
        start = (np.cumsum(nums) - np.asarray(nums)).astype(np.int32)
        keep = (rs.rand(I, B, NC, kmax) < dens).astype(np.uint8)
        for b, n in enumerate(nums):
            keep[:, b, :, n:] = 0                                                       # (slots past a clip's tubes are never set by step_detect_nms)
        boxes = [(rs.rand(N, 4) * 400).astype(np.float32) for _ in range(I)]
        scores = [rs.rand(N, NC + 3).astype(np.float32) for _ in range(I)]              # (row stride > NC)
        cap = NC * kmax
        dk, ds = bk.dev(keep), bk.dev(start)
        db, dsx = [bk.dev(b_) for b_ in boxes], [bk.dev(s_) for s_ in scores]
        ob, os_ = bk.dev(np.full((I * B * cap, 4), -1, np.float32)), bk.dev(np.full((I * B * cap,), -1, np.float32))
        oc, ot = bk.dev(np.full((I * B * cap,), -1, np.int64)), bk.dev(np.full((I * B * cap,), -1, np.int64))
        cnt = bk.dev(np.full((I * B,), -1, np.int32))
        bp = (ctypes.c_void_p * I)(*[int(b_.ptr.value if hasattr(b_.ptr, "value") else b_.ptr) for b_ in db])
        sp = (ctypes.c_void_p * I)(*[int(s_.ptr.value if hasattr(s_.ptr, "value") else s_.ptr) for s_ in dsx])
        ss = (ctypes.c_longlong * I)(*[NC + 3] * I)
        rc = bk.lib.step_detect_compact(dk.ptr, ctypes.cast(bp, ctypes.c_void_p), ctypes.cast(sp, ctypes.c_void_p), ctypes.cast(ss, ctypes.c_void_p), ds.ptr,
                                        I, B, NC, kmax, 400.0, 300.0, ob.ptr, os_.ptr, oc.ptr, ot.ptr, cnt.ptr, bk.stream)
        assert rc == 0, rc
        gb, gs, gc, gt, gn = ob.get(), os_.get(), oc.get(), ot.get(), cnt.get()
        wh = np.asarray([400.0, 300.0, 400.0, 300.0], np.float32)
        for i in range(I):
            for b in range(B):
                g = i * B + b
                c_, j_ = np.nonzero(keep[i, b])
                assert gn[g] == len(c_), (I, g, gn[g], len(c_))
                sl = slice(g * cap, g * cap + len(c_))
                assert np.array_equal(gc[sl], c_) and np.array_equal(gt[sl], j_)
                assert np.array_equal(gb[sl], boxes[i][start[b] + j_] / wh)
                assert np.array_equal(gs[sl], scores[i][start[b] + j_, c_])
                assert (gc[g * cap + len(c_):(g + 1) * cap] == -1).all()               # nothing written past the group's rows
    # no classes: only the zero counts are written; too many iterations are refused
    cnt = bk.dev(np.full((4,), -1, np.int32))
    assert bk.lib.step_detect_compact(None, None, None, None, None, 2, 2, 0, 5, 1.0, 1.0, None, None, None, None, cnt.ptr, bk.stream) == 0
    assert (cnt.get() == 0).all()
    assert bk.lib.step_detect_compact(None, None, None, None, None, 9, 2, 4, 5, 1.0, 1.0, None, None, None, None, cnt.ptr, bk.stream) == -2


def case_pack_weight_perm_folds_flatten_order(bk, golden):
    # Linear over an NCHW-flattened feature (c*HW+hw) evaluated on an NHWC-flattened one (hw*C+c)
    rs = np.random.RandomState(12)
    C, HW, O, M = 8, 4, 4, 5
    feat = rs.randn(M, C, HW).astype(np.float32)
    wl = rs.randn(O, C * HW).astype(np.float32)
    ref = feat.reshape(M, -1) @ wl.T
    perm = np.array([(j % C) * HW + (j // C) for j in range(C * HW)], np.int32)   # packed channel j = hw*C+c
    wp = pack_weight(bk, wl.reshape(O, C * HW, 1, 1, 1), F32, perm)
    x = bk.dev(np.ascontiguousarray(np.transpose(feat, (0, 2, 1))).reshape(M, 1, 1, 1, C * HW))
    y = bk.dev(np.zeros((M, 1, 1, 1, O), np.float32))
    d = _capi.ConvDesc(dtype=0, N=M, D=1, H=1, W=1, Cin=C * HW, Cout=O, kd=1, kh=1, kw=1, x_cstride=C * HW, x_coff=0,
                       y_cstride=O, y_coff=0, res_cstride=0, res_coff=0, relu=0, split=0, y2_cstride=0, y2_coff=0)
    assert bk.lib.step_conv_forward(ctypes.byref(d), x.ptr, wp.ptr, None, None, None, y.ptr, None, bk.stream) == 0
    assert np.abs(y.get().reshape(M, O) - ref).max() < 1e-5


def _stem_case(bk, shape, dts, Cout):
    N, T, H, W = shape
    rs = np.random.RandomState(T + H)
    x = rs.uniform(-1, 1, (N, T, 3, H, W)).astype(np.float32)
    w = (rs.randn(Cout, 3, 7, 7, 7) / np.sqrt(1029)).astype(np.float32)
    scale = (1 + 0.1 * rs.randn(Cout)).astype(np.float32)
    shift = (0.2 * rs.randn(Cout)).astype(np.float32)
    for dt in dts:
        got = run_stem(bk, x, w, scale, shift, dt)
        ref = ref_stem(x, w, scale, shift, dt)
        assert got.shape == ref.shape
        err = np.abs(got - ref).max() / np.abs(ref).max()
        assert err < tol(dt), (shape, dt, err)


def case_stem(bk, golden):
    _stem_case(bk, (1, 8, 32, 32), (F32, BF16), 64)
    _stem_case(bk, (2, 5, 18, 22), (F32, BF16), 40)     # odd T, W % 4 != 0 -> scalar staging path
    _stem_case(bk, (1, 4, 17, 19), (F32, F16), 40)


def case_stem_golden(bk, golden):
    g = golden("ops_golden")
    shapes = {"conv3d.weight": (16, 3, 7, 7, 7)}
    for nme in ("weight", "bias", "running_mean", "running_var"):
        shapes["batch3d." + nme] = (16,)
    sd = R.fill_state_dict(shapes, "golden.unit.stem.")
    scale = (sd["batch3d.weight"] / torch.sqrt(sd["batch3d.running_var"] + 1e-5)).numpy()
    shift = (sd["batch3d.bias"] - sd["batch3d.running_mean"] * torch.from_numpy(scale)).numpy()
    x = R.fill_tensor("golden.unit.stem.in", (1, 3, 9, 21, 19), "image").permute(0, 2, 1, 3, 4).contiguous().numpy()
    got = run_stem(bk, x, sd["conv3d.weight"].numpy(), scale, shift, F32)
    ref = g["unit_stem_out"]
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= 2e-5 * np.abs(ref).max()


ALL = sorted(k for k in globals() if k.startswith("case_"))


# ------------------------------------------------------------------ larger, GPU-only cases
def big_conv_shapes(bk, golden):
    """Real I3D layer shapes (C2 geometry, one clip) -- too slow for the interpreter."""
    for case in ((1, 64, 192, 4, 56, 56, (3, 3, 3)), (1, 192, 96, 4, 28, 28, (1, 1, 1)), (1, 128, 256, 8, 14, 14, (3, 3, 3)),
                 (2, 832, 384, 3, 7, 7, (1, 1, 1)), (9, 256, 256, 1, 7, 7, (1, 3, 3)), (1, 160, 320, 9, 25, 25, (3, 3, 3))):
        _conv_case(bk, case, (F32, BF16, F16))


def big_stem(bk, golden):
    _stem_case(bk, (1, 16, 112, 112), (F32, BF16), 64)
    _stem_case(bk, (1, 8, 224, 224), (BF16,), 64)


def big_roi(bk, golden):
    """AVA-shaped ROIAlign: 99 rois on [9,832,25,25], fp32, bit-exact against the C oracle."""
    rs = np.random.RandomState(21)
    feat = np.maximum(rs.randn(9, 832, 25, 25), 0).astype(np.float32)
    a = R.anchors()[:11] * 400.0
    rois = np.concatenate([np.repeat(np.arange(9), 11)[:, None].astype(np.float32),
                           np.tile(a, (9, 1)) + rs.uniform(-5, 5, (99, 4)).astype(np.float32)], 1).astype(np.float32)
    ref = oracle.roi_align_forward(feat, rois, (7, 7), 1 / 16., 0)
    for layout in (NCHW, NHWC):
        assert np.array_equal(run_roi_align(bk, feat, rois, (7, 7), 1 / 16., 0, layout), ref)


def big_nms(bk, golden):
    rs = np.random.RandomState(22)
    for G, kmax in ((480, 34), (60, 109), (3, 1000)):
        xy = rs.uniform(0, 300, (G, kmax, 2))
        wh = rs.uniform(10, 150, (G, kmax, 2))
        boxes = np.concatenate([xy, xy + wh], 2).astype(np.float32)
        scores = rs.uniform(0, 1, (G, kmax)).astype(np.float32)
        counts = rs.randint(kmax // 2, kmax + 1, G).astype(np.int32)
        assert np.array_equal(run_nms(bk, boxes, scores, counts, 0.4), oracle.nms_batched(boxes, scores, counts, 0.4))


def big_wgrad_full_size_properties(bk, golden):
    """Weight gradients at the C4 layer sizes (AVA clip: conv3d_2c on 18x100x100, a 25x25 Inception conv, the heads' convs on
    7x7 maps), where no CPU oracle finishes in seconds -- size-independent properties instead:
      * the job split does not matter: one-row jobs (option wgrad_minpix = 16), the default and one job per tile
        (wgrad_minpix = 10^6) agree to fp32 summation-order noise;
      * linearity in dy: wgrad(x, a + b) == wgrad(x, a) + wgrad(x, b);
      * a checksum against a dense contraction: sum over taps and input channels of dw[co] for an all-ones x equals the
        (border-clipped) tap counts times sum(dy[co]) -- evaluated in closed form."""
    import torch
    torch.manual_seed(8)
    for (N, Cin, Cout, D, H, W, k) in ((1, 64, 192, 18, 100, 100, (3, 3, 3)), (1, 160, 320, 9, 25, 25, (3, 3, 3)),
                                       (15, 256, 256, 1, 7, 7, (1, 3, 3)), (5, 192, 384, 3, 7, 7, (3, 3, 3)), (45, 256, 1024, 1, 7, 7, (1, 1, 1))):
        x = torch.randn(N, D, H, W, Cin).numpy()
        a = torch.randn(N, D, H, W, Cout).numpy()
        b = torch.randn(N, D, H, W, Cout).numpy()
        xd = bk.dev(x)
        d = _capi.ConvDesc(dtype=F32, N=N, D=D, H=H, W=W, Cin=Cin, Cout=Cout, kd=k[0], kh=k[1], kw=k[2], x_cstride=Cin, x_coff=0,
                           y_cstride=Cout, y_coff=0, res_cstride=0, res_coff=0, relu=0, split=0, y2_cstride=0, y2_coff=0)

        def wg(xbuf, dy, minpix=0):
            with _capi.options(bk.lib, wgrad_minpix=minpix):
                dw = bk.dev(np.zeros((Cout, Cin) + k, np.float32))
                assert bk.lib.step_conv_wgrad(ctypes.byref(d), xbuf.ptr, bk.dev(dy).ptr, dw.ptr, 0, bk.stream) == 0
                return dw.get()

        base = wg(xd, a)
        scale = np.abs(base).max()
        # one job per tile = one fp32 chain over every pixel of the map (180 000 terms for conv3d_2c): its rounding noise
        # grows with the chain length, hence the wider bound there
        for mp, bound in ((16, 2e-5), (1000000, 1e-4)):
            assert np.abs(wg(xd, a, mp) - base).max() < bound * scale, (Cin, Cout, k, mp)
        assert np.abs(wg(xd, a + b) - (base + wg(xd, b))).max() < 2e-5 * scale, (Cin, Cout, k)
        ones = bk.dev(np.ones((N, D, H, W, Cin), np.float32))
        got = wg(ones, a).astype(np.float64)
        # dw[co, ci, t] = sum of dy[co] over the output pixels whose tap t lands inside the map
        dy = a.astype(np.float64)
        for t in np.ndindex(*k):
            o = [t[i] - k[i] // 2 for i in range(3)]
            sl = tuple(slice(max(0, -o[i]), (D, H, W)[i] - max(0, o[i])) for i in range(3))
            want = dy[(slice(None),) + sl].sum(axis=(0, 1, 2, 3))
            assert np.abs(got[(slice(None), 0) + t] - want).max() < 1e-4 * max(1.0, np.abs(want).max()), (Cin, Cout, k, t)


def big_adam_full_size(bk, golden):
    """The fused Adam at the C4 parameter count (44.4 M fp32) against torch's own Adam on the same device:
    2 steps, per-segment lr / weight_decay, 2e-6 of max|p|."""
    import torch
    torch.manual_seed(3)
    sizes = [4_000_000 + 64 * i for i in range(11)]                 # multiples of 4
    n = sum(sizes)
    assert abs(n - 44_422_936) < 500_000
    ps = [torch.nn.Parameter(torch.randn(s_, device="cuda") * 0.05) for s_ in sizes]
    lrs = [1e-3 * (1 + i % 3) for i in range(len(sizes))]
    wds = [0.0 if i % 2 else 1e-4 for i in range(len(sizes))]
    opt = torch.optim.Adam([{"params": [p], "lr": lr, "weight_decay": wd} for p, lr, wd in zip(ps, lrs, wds)], lr=1e-3)
    P = torch.cat([p.detach().reshape(-1) for p in ps]).clone()
    M, V = torch.zeros_like(P), torch.zeros_like(P)
    ends = torch.tensor(np.cumsum(sizes), dtype=torch.int64, device="cuda")
    LR, WD = torch.tensor(lrs, device="cuda"), torch.tensor(wds, device="cuda")
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    for step_no in (1, 2):
        G = torch.randn(n, device="cuda")
        off = 0
        for p, s_ in zip(ps, sizes):
            p.grad = G[off:off + s_].clone()
            off += s_
        opt.step()
        assert bk.lib.step_adam_flat(vp(P), vp(G), vp(M), vp(V), n, vp(ends), vp(LR), vp(WD), len(sizes), 0.9, 0.999, 1e-8, step_no, 1.0, 1,
                                     bk.stream) == 0
        ref = torch.cat([p.detach().reshape(-1) for p in ps])
        assert float((P - ref).abs().max()) <= 2e-6 * float(ref.abs().max()), step_no
        assert float(G.abs().max()) == 0.0


def big_pre_pool_ava(bk, golden):
    """step_conv_forward_pre_pool at the AVA shape (100 x 100 x 18-plane maps): the planner alone takes a general box there, the pooled call
    re-plans onto 4 x 8 x 8 tiles (845 against 720, within its 25 % allowance; tools/prepool_ab.py: 521 against 571 us) with partial tiles on
    both sides (100 = 12.5 x 8) and a partial plane group (18 = 4.5 x 4) -- BIT-IDENTICAL to step_conv_forward_pre + step_maxpool3d_tf."""
    rs = np.random.RandomState(63)
    dt, (N, D, H, W), Cout = BF16, (2, 18, 100, 100), 192
    info = (ctypes.c_int * 10)()
    x = rs.randn(N, 64, D, H, W).astype(np.float32)
    wa = (rs.randn(64, 64, 1, 1, 1) / 8).astype(np.float32)
    wb = (rs.randn(Cout, 64, 3, 3, 3) / np.sqrt(64 * 27)).astype(np.float32)
    sa, ha = (1 + 0.1 * rs.randn(64)).astype(np.float32), (0.2 * rs.randn(64)).astype(np.float32)
    sb, hb = (1 + 0.1 * rs.randn(Cout)).astype(np.float32), (0.2 * rs.randn(Cout)).astype(np.float32)
    xe = bk.dev(encode(cl(x), dt))
    wpa, wpb = pack_weight(bk, wa, dt), pack_weight(bk, wb, dt)
    dsa, dha, dsb, dhb = bk.dev(sa), bk.dev(ha), bk.dev(sb), bk.dev(hb)
    Hp, Wp = bk.lib.step_pool_out_size(H, 3, 2), bk.lib.step_pool_out_size(W, 3, 2)
    db = _capi.ConvDesc(dtype=dt, N=N, D=D, H=H, W=W, Cin=64, Cout=Cout, kd=3, kh=3, kw=3, x_cstride=64, x_coff=0, y_cstride=Cout, y_coff=0,
                        res_cstride=0, res_coff=0, relu=1, split=0, y2_cstride=0, y2_coff=0)
    assert bk.lib.step_conv_plan_info(ctypes.byref(db), info, 10) == 0 and info[1] == 0          # the planner's own choice: a general box
    full = bk.dev(np.zeros((N, D, H, W, Cout), NP_DT[dt]))
    assert bk.lib.step_conv_forward_pre(ctypes.byref(db), xe.ptr, wpb.ptr, dsb.ptr, dhb.ptr, wpa.ptr, dsa.ptr, dha.ptr, 64, full.ptr, bk.stream) == 0
    want = bk.dev(np.zeros((N, D, Hp, Wp, Cout), NP_DT[dt]))
    assert bk.lib.step_maxpool3d_tf(dt, full.ptr, N, D, H, W, Cout, Cout, 0, 1, 3, 3, 1, 2, 2, want.ptr, Cout, 0, bk.stream) == 0
    nb = bk.lib.step_conv_pre_pool_workspace_bytes(ctypes.byref(db))
    assert nb > 0
    ws = bk.dev(np.full(nb // 2, 0x7f7f, np.uint16))
    got = bk.dev(np.full((N, D, Hp, Wp, Cout), 3, NP_DT[dt]))
    assert bk.lib.step_conv_forward_pre_pool(ctypes.byref(db), xe.ptr, wpb.ptr, dsb.ptr, dhb.ptr, wpa.ptr, dsa.ptr, dha.ptr, 64, got.ptr, ws.ptr, nb, bk.stream) == 0
    g_, w_ = got.get(), want.get()
    assert np.array_equal(g_, w_), int((g_ != w_).sum())


GPU_ONLY = ["big_pre_pool_ava", "big_conv_shapes", "big_stem", "big_roi", "big_nms", "big_wgrad_full_size_properties", "big_adam_full_size"]
