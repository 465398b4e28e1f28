"""Kernel SOURCE (step_amd/csrc/*.hip) executed on the host SIMT interpreter (tests/emul) and
checked against the oracle.  This validates index logic / tiling / arithmetic order on the CPU-only
build container; the same checks run against the real gfx950 build in tests/test_gpu_*.py."""
import ctypes

import numpy as np
import pytest
import torch

import oracle
from oracle import i3d_ref as R
from step_amd import _capi
from tests.emul import emul_lib as E


def nhwc(x):
    return np.ascontiguousarray(np.transpose(x, (0, 2, 3, 1)))


def nchw(x):
    return np.ascontiguousarray(np.transpose(x, (0, 3, 1, 2)))


def run_roi_align(feat_nchw, rois, pooled, scale, sr, layout, dt=_capi.F32):
    L = E.lib()
    B, C, H, W = feat_nchw.shape
    K = rois.shape[0]
    ph, pw = pooled
    f = feat_nchw if layout == _capi.NCHW else nhwc(feat_nchw)
    fe = E.encode(f, dt)
    oshape = (K, C, ph, pw) if layout == _capi.NCHW else (K, ph, pw, C)
    out = np.zeros(oshape, E.NP_DT[dt])
    rc = L.step_roi_align_forward(E.ptr(fe), dt, layout, E.ptr(rois), K, B, C, H, W, ph, pw, scale, sr, E.ptr(out), None)
    assert rc == 0
    o = E.decode(out, dt)
    return o if layout == _capi.NCHW else nchw(o)


@pytest.mark.parametrize("layout", [_capi.NCHW, _capi.NHWC])
def test_roi_align_forward_golden_bit_exact(golden, layout):
    g = golden("roi_nms_golden")
    feat = R.fill_tensor("golden.roi.feat", (3, 6, 25, 25), "image").numpy()   # C=6: scalar-lane path
    for tag, pooled, sr in (("p7s0", (7, 7), 0), ("p7s2", (7, 7), 2), ("p3x5s0", (3, 5), 0)):
        out = run_roi_align(feat, g["align_rois"], pooled, 1 / 16., sr, layout)
        assert np.array_equal(out, g["align_out_" + tag]), tag


def test_roi_align_forward_vector_path_and_tubes(golden):
    g = golden("roi_nms_golden")
    conv = R.fill_tensor("golden.roi.conv", (2, 3, 16, 25, 25), "feat").numpy().reshape(6, 16, 25, 25)
    out = run_roi_align(conv, g["tube_rois"].reshape(-1, 5), (7, 7), 1 / 16., 0, _capi.NHWC)   # C=16: 16-byte lanes
    assert np.array_equal(out, g["tube_out"])


def test_roi_align_forward_bf16():
    rs = np.random.RandomState(1)
    x = rs.randn(2, 16, 13, 11).astype(np.float32)
    rois = np.array([[0, 3, 5, 150, 120], [1, 20, 30, 60.5, 99.25]], np.float32)
    xq = E.quantize(x, _capi.BF16)
    ref = oracle.roi_align_forward(xq, rois, (7, 7), 1 / 16., 0)
    out = run_roi_align(x, rois, (7, 7), 1 / 16., 0, _capi.NHWC, _capi.BF16)
    assert np.abs(out - ref).max() <= 2 ** -8 * np.abs(ref).max()      # one bf16 rounding of the output


@pytest.mark.parametrize("layout", [_capi.NCHW, _capi.NHWC])
def test_roi_align_backward(layout):
    L = E.lib()
    rs = np.random.RandomState(2)
    B, C, H, W = 2, 8, 9, 12
    rois = np.array([[0, 0, 0, 190, 140], [1, 33.3, 20.1, 120.7, 100.2], [1, -20, 100, 90, 250], [0, 50, 50, 50.5, 50.5]], np.float32)
    K = rois.shape[0]
    for sr in (0, 2):
        g = rs.randn(K, C, 7, 7).astype(np.float32)
        ref = oracle.roi_align_backward(g, rois, (7, 7), 1 / 16., sr, (B, C, H, W))
        gi = np.full((B, C, H, W) if layout == _capi.NCHW else (B, H, W, C), 7.0, np.float32)   # must be zeroed by the op
        gg = g if layout == _capi.NCHW else nhwc(g)
        rc = L.step_roi_align_backward(E.ptr(gg), layout, E.ptr(rois), K, B, C, H, W, 7, 7, 1 / 16., sr, E.ptr(gi), None)
        assert rc == 0
        got = gi if layout == _capi.NCHW else nchw(gi)
        assert np.abs(got - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())   # atomics: order differs


@pytest.mark.parametrize("layout", [_capi.NCHW, _capi.NHWC])
def test_roi_pool_forward_backward(layout):
    L = E.lib()
    rs = np.random.RandomState(3)
    B, C, H, W = 2, 5, 10, 13
    x = rs.randn(B, C, H, W).astype(np.float32)
    rois = np.array([[0, 0, 0, 200, 150], [1, 17, 9, 88, 140], [0, 300, 300, 400, 400], [1, 40, 40, 41, 41], [0, -30, -10, 50, 60]], np.float32)
    K = rois.shape[0]
    ref, refarg = oracle.roi_pool_forward(x, rois, (7, 7), 1 / 16.)
    xx = x if layout == _capi.NCHW else nhwc(x)
    oshape = (K, C, 7, 7) if layout == _capi.NCHW else (K, 7, 7, C)
    out = np.zeros(oshape, np.float32)
    arg = np.zeros(oshape, np.int32)
    assert L.step_roi_pool_forward(E.ptr(xx), _capi.F32, layout, E.ptr(rois), K, B, C, H, W, 7, 7, 1 / 16., E.ptr(out), E.ptr(arg), None) == 0
    o, a = (out, arg) if layout == _capi.NCHW else (nchw(out), nchw(arg))
    assert np.array_equal(o, ref) and np.array_equal(a, refarg)
    g = rs.randn(K, C, 7, 7).astype(np.float32)
    refg = oracle.roi_pool_backward(g, refarg, rois, (7, 7), x.shape)
    gg = g if layout == _capi.NCHW else nhwc(g)
    gi = np.full(xx.shape, 3.0, np.float32)
    assert L.step_roi_pool_backward(E.ptr(gg), E.ptr(arg), layout, E.ptr(rois), K, B, C, H, W, 7, 7, E.ptr(gi), None) == 0
    got = gi if layout == _capi.NCHW else nchw(gi)
    assert np.abs(got - refg).max() <= 1e-5


def test_roi_empty_and_bad_args():
    L = E.lib()
    x = np.zeros((1, 4, 4, 8), np.float32)
    assert L.step_roi_align_forward(E.ptr(x), 0, 1, None, 0, 1, 8, 4, 4, 7, 7, 1.0, 0, None, None) == 0      # K = 0
    assert L.step_roi_align_forward(E.ptr(x), 9, 1, E.ptr(x), 1, 1, 8, 4, 4, 7, 7, 1.0, 0, E.ptr(x), None) == -1  # dtype
    assert L.step_roi_align_forward(None, 0, 1, E.ptr(x), 1, 1, 8, 4, 4, 7, 7, 1.0, 0, E.ptr(x), None) == -3   # null
    assert L.step_roi_align_forward(E.ptr(x), 0, 5, E.ptr(x), 1, 1, 8, 4, 4, 7, 7, 1.0, 0, E.ptr(x), None) == -4  # layout


def run_nms(boxes, scores, counts, thr):
    L = E.lib()
    G, kmax = scores.shape
    keep = np.full((G, kmax), 9, np.uint8)
    nb = L.step_nms_scratch_bytes(G, kmax)
    scratch = np.zeros(max(nb, 1), np.uint8)
    rc = L.step_nms_batched(E.ptr(boxes), E.ptr(scores), E.ptr(counts), G, kmax, thr, E.ptr(keep), E.ptr(scratch) if nb else None, None)
    assert rc == 0
    return keep


def test_nms_golden_bit_exact(golden):
    g = golden("roi_nms_golden")
    for i in range(int(g["nms_count"])):
        b, s, thr = g["nms%d_boxes" % i], g["nms%d_scores" % i], float(g["nms%d_thr" % i])
        n = b.shape[0]
        keep = run_nms(b[None].copy(), s[None].copy(), np.array([n], np.int32), thr)
        assert np.array_equal(np.nonzero(keep[0])[0], g["nms%d_keep" % i]), i


@pytest.mark.parametrize("kmax", [11, 34, 64, 109])
def test_nms_batched_groups_and_ties(kmax):
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
    keep = run_nms(boxes, scores, counts, 0.4)
    assert np.array_equal(keep, oracle.nms_batched(boxes, scores, counts, 0.4))


POOLS = [((1, 3, 3), (1, 2, 2)), ((3, 3, 3), (2, 2, 2)), ((3, 3, 3), (1, 1, 1)), ((2, 2, 2), (2, 2, 2))]


@pytest.mark.parametrize("k,s", POOLS)
@pytest.mark.parametrize("dt", [_capi.F32, _capi.BF16])
def test_maxpool_tf(k, s, dt):
    L = E.lib()
    rs = np.random.RandomState(5)
    N, C, D, H, W = 2, 16, 5, 9, 7                    # odd sizes: ceil-mode overhang, negative values
    x = rs.randn(N, C, D, H, W).astype(np.float32)
    xq = E.quantize(x, dt)
    ref = R.maxpool_tf(torch.from_numpy(xq), k, s).numpy()
    Do, Ho, Wo = (L.step_pool_out_size(a, b, c) for a, b, c in zip((D, H, W), k, s))
    assert ref.shape == (N, C, Do, Ho, Wo)
    xcl = E.encode(np.ascontiguousarray(np.transpose(x, (0, 2, 3, 4, 1))), dt)
    # write into a channel slice of a wider buffer
    ycs, yoff = 40, 8
    y = np.zeros((N, Do, Ho, Wo, ycs), E.NP_DT[dt])
    rc = L.step_maxpool3d_tf(dt, E.ptr(xcl), N, D, H, W, C, C, 0, k[0], k[1], k[2], s[0], s[1], s[2], E.ptr(y), ycs, yoff, None)
    assert rc == 0
    got = np.transpose(E.decode(y, dt)[..., yoff:yoff + C], (0, 4, 1, 2, 3))
    assert np.array_equal(got, ref)
    assert not E.decode(y, dt)[..., :yoff].any() and not E.decode(y, dt)[..., yoff + C:].any()


def test_maxpool_zero_pad_value():
    L = E.lib()
    x = -np.ones((1, 4, 5, 5, 4), np.float32)
    y = np.zeros((1, 2, 3, 3, 4), np.float32)
    assert L.step_maxpool3d_tf(0, E.ptr(x), 1, 4, 5, 5, 4, 4, 0, 3, 3, 3, 2, 2, 2, E.ptr(y), 4, 0, None) == 0
    ref = R.maxpool_tf(-torch.ones(1, 4, 4, 5, 5), (3, 3, 3), (2, 2, 2)).numpy()
    assert np.array_equal(np.transpose(y, (0, 4, 1, 2, 3)), ref) and y.max() == 0.0 and y.min() == -1.0


def test_avgpool_hw():
    L = E.lib()
    rs = np.random.RandomState(6)
    x = rs.randn(2, 8, 3, 13, 13).astype(np.float32)
    ref = torch.nn.functional.avg_pool3d(torch.from_numpy(x), (1, 13, 13), (1, 1, 1)).numpy()
    xcl = np.ascontiguousarray(np.transpose(x, (0, 2, 3, 4, 1)))
    y = np.zeros((2, 3, 1, 1, 8), np.float32)
    assert L.step_avgpool_hw(0, E.ptr(xcl), 2, 3, 13, 13, 8, 13, 13, E.ptr(y), None) == 0
    assert np.abs(np.transpose(y, (0, 4, 1, 2, 3)) - ref).max() < 1e-6


def test_transpose_cs():
    L = E.lib()
    rs = np.random.RandomState(7)
    x = rs.randn(2, 37, 50).astype(np.float32)
    y = np.zeros((2, 50, 37), np.float32)
    assert L.step_transpose_cs(E.ptr(x), 0, E.ptr(y), 0, 2, 37, 50, 1, None) == 0
    assert np.array_equal(y, np.transpose(x, (0, 2, 1)))
    z = np.zeros_like(x)
    assert L.step_transpose_cs(E.ptr(y), 0, E.ptr(z), 0, 2, 37, 50, 0, None) == 0
    assert np.array_equal(z, x)
    yb = np.zeros((2, 50, 37), np.uint16)
    assert L.step_transpose_cs(E.ptr(x), 0, E.ptr(yb), 1, 2, 37, 50, 1, None) == 0
    assert np.array_equal(yb, E.to_bf16_bits(np.transpose(x, (0, 2, 1))))
