"""conv_igemm.hip on the host SIMT interpreter vs torch fp32 conv (the oracle's arithmetic library)."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from step_amd import _capi
from tests.emul import emul_lib as E


def cl(x):   # NCDHW -> NDHWC
    return np.ascontiguousarray(np.transpose(x, (0, 2, 3, 4, 1)))


def uncl(x):
    return np.ascontiguousarray(np.transpose(x, (0, 4, 1, 2, 3)))


def pack_weight(w, dt, perm=None):
    L = E.lib()
    Cout, Cin, kd, kh, kw = w.shape
    n = L.step_conv_packed_elems(Cout, Cin, kd, kh, kw)
    out = np.zeros(n, E.NP_DT[dt])
    w = np.ascontiguousarray(w, np.float32)
    p = None if perm is None else np.ascontiguousarray(perm, np.int32)
    assert L.step_conv_pack_weight(E.ptr(w), Cout, Cin, kd, kh, kw, dt, E.ptr(p), E.ptr(out), None) == 0
    return out


def run_conv(x, w, scale, shift, dt, relu=True, res=None, x_pad=(0, 0), y_pad=(0, 0)):
    """x NCDHW fp32, w torch layout.  x_pad/y_pad = (extra channels before, after) in the buffers."""
    L = E.lib()
    N, Cin, D, H, W = x.shape
    Cout = w.shape[0]
    xc = cl(x)
    xb = np.zeros(xc.shape[:-1] + (x_pad[0] + Cin + x_pad[1],), np.float32)
    xb[..., x_pad[0]:x_pad[0] + Cin] = xc
    xb[..., :x_pad[0]] = 77.0           # must never be read
    xb[..., x_pad[0] + Cin:] = -55.0
    xe = E.encode(xb, dt)
    yb = np.zeros((N, D, H, W, y_pad[0] + Cout + y_pad[1]), E.NP_DT[dt])
    wp = pack_weight(w, dt)
    d = _capi.ConvDesc(dtype=dt, N=N, D=D, H=H, W=W, Cin=Cin, Cout=Cout, kd=w.shape[2], kh=w.shape[3], kw=w.shape[4],
                       x_cstride=xb.shape[-1], x_coff=x_pad[0], y_cstride=yb.shape[-1], y_coff=y_pad[0],
                       res_cstride=Cout, res_coff=0, relu=int(relu))
    re = None if res is None else E.encode(cl(res), dt)
    sc = None if scale is None else np.ascontiguousarray(scale, np.float32)
    sh = None if shift is None else np.ascontiguousarray(shift, np.float32)
    rc = L.step_conv_forward(ctypes.byref(d), E.ptr(xe), E.ptr(wp), E.ptr(sc), E.ptr(sh), E.ptr(re), E.ptr(yb), None)
    assert rc == 0, rc
    y = E.decode(yb, dt)
    assert not y[..., :y_pad[0]].any() and not y[..., y_pad[0] + Cout:].any()
    return uncl(y[..., y_pad[0]:y_pad[0] + Cout])


def ref_conv(x, w, scale, shift, dt, relu=True, res=None):
    xq = torch.from_numpy(E.quantize(x, dt))
    wq = torch.from_numpy(E.quantize(w, dt))
    pad = tuple(k // 2 for k in w.shape[2:])
    y = F.conv3d(xq, wq, padding=pad)
    if scale is not None:
        y = y * torch.from_numpy(scale).view(1, -1, 1, 1, 1)
    if shift is not None:
        y = y + torch.from_numpy(shift).view(1, -1, 1, 1, 1)
    if res is not None:
        y = y + torch.from_numpy(E.quantize(res, dt))
    if relu:
        y = F.relu(y)
    return y.numpy()


def tol(dt):
    return {_capi.F32: 2e-5, _capi.BF16: 2 ** -7, _capi.F16: 2 ** -9}[dt]


CASES = [
    # (N, Cin, Cout, D, H, W, kernel)
    (1, 16, 32, 2, 8, 16, (3, 3, 3)),       # exactly one 8x16 tile, one slab (half empty), one n-block
    (2, 24, 48, 3, 9, 7, (3, 3, 3)),        # odd sizes, Cin/Cout not multiples of 32
    (1, 40, 72, 2, 5, 37, (3, 3, 3)),       # wide tile shape (4x32), 2 slabs, 3 n-blocks
    (1, 96, 208, 1, 14, 14, (3, 3, 3)),     # an I3D 4b shape: 7 n-blocks
    (2, 32, 40, 3, 7, 7, (1, 3, 3)),        # 2-D 3x3 conv (frames on D)
    (2, 72, 100, 2, 5, 9, (1, 1, 1)),       # pointwise: flat tiles, tail tile
    (1, 200, 16, 1, 13, 13, (1, 1, 1)),     # 7 slabs
]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("dt", [_capi.F32, _capi.BF16])
def test_conv_unit(case, dt):
    N, Cin, Cout, D, H, W, k = case
    rs = np.random.RandomState(Cin * 7 + Cout)
    x = rs.randn(N, Cin, D, H, W).astype(np.float32)
    w = (rs.randn(Cout, Cin, *k) / np.sqrt(Cin * np.prod(k))).astype(np.float32)
    scale = (1 + 0.1 * rs.randn(Cout)).astype(np.float32)
    shift = (0.2 * rs.randn(Cout)).astype(np.float32)
    got = run_conv(x, w, scale, shift, dt, x_pad=(8, 8), y_pad=(16, 8))
    ref = ref_conv(x, w, scale, shift, dt)
    err = np.abs(got - ref).max() / np.abs(ref).max()
    assert err < tol(dt), err


def test_conv_residual_norelu_f16_and_bias_only():
    rs = np.random.RandomState(11)
    x = rs.randn(2, 32, 1, 7, 7).astype(np.float32)
    w = (rs.randn(64, 32, 1, 1, 1) / 6).astype(np.float32)
    res = rs.randn(2, 64, 1, 7, 7).astype(np.float32)
    got = run_conv(x, w, None, None, _capi.F32, relu=True, res=res)
    assert np.abs(got - ref_conv(x, w, None, None, _capi.F32, True, res)).max() < 1e-5
    bias = rs.randn(64).astype(np.float32)
    got = run_conv(x, w, None, bias, _capi.F16, relu=False)
    ref = ref_conv(x, w, None, bias, _capi.F16, False)
    assert np.abs(got - ref).max() / np.abs(ref).max() < tol(_capi.F16)


def test_pack_weight_perm_folds_flatten_order():
    # Linear over an NCHW-flattened feature (c*49+hw) evaluated on an NHWC-flattened one (hw*C+c)
    rs = np.random.RandomState(12)
    C, HW, O, M = 8, 4, 4, 5
    feat = rs.randn(M, C, HW).astype(np.float32)
    wl = rs.randn(O, C * HW).astype(np.float32)
    ref = feat.reshape(M, -1) @ wl.T
    perm = np.array([(j % C) * HW + (j // C) for j in range(C * HW)], np.int32)   # packed channel j = hw*C+c
    L = E.lib()
    n = L.step_conv_packed_elems(O, C * HW, 1, 1, 1)
    wp = np.zeros(n, np.float32)
    assert L.step_conv_pack_weight(E.ptr(wl), O, C * HW, 1, 1, 1, 0, E.ptr(perm), E.ptr(wp), None) == 0
    x = np.ascontiguousarray(np.transpose(feat, (0, 2, 1))).reshape(M, 1, 1, 1, C * HW)
    y = np.zeros((M, 1, 1, 1, O), np.float32)
    d = _capi.ConvDesc(dtype=0, N=M, D=1, H=1, W=1, Cin=C * HW, Cout=O, kd=1, kh=1, kw=1, x_cstride=C * HW, x_coff=0,
                       y_cstride=O, y_coff=0, res_cstride=0, res_coff=0, relu=0)
    assert L.step_conv_forward(ctypes.byref(d), E.ptr(x), E.ptr(wp), None, None, None, E.ptr(y), None) == 0
    assert np.abs(y.reshape(M, O) - ref).max() < 1e-5


def run_stem(x_ntchw, w, scale, shift, dt):
    L = E.lib()
    N, T, _, H, W = x_ntchw.shape
    Cout = w.shape[0]
    n = L.step_stem_packed_elems(Cout)
    wp = np.zeros(n, E.NP_DT[dt])
    w = np.ascontiguousarray(w, np.float32)
    assert L.step_stem_pack_weight(E.ptr(w), Cout, dt, E.ptr(wp), None) == 0
    To, Ho, Wo = (T - 2) // 2 + 1, (H - 2) // 2 + 1, (W - 2) // 2 + 1
    y = np.zeros((N, To, Ho, Wo, Cout), E.NP_DT[dt])
    xe = E.encode(x_ntchw, dt)
    rc = L.step_stem_forward(dt, E.ptr(xe), N, T, H, W, E.ptr(wp), E.ptr(scale), E.ptr(shift), Cout, E.ptr(y), Cout, 0, None)
    assert rc == 0
    return uncl(E.decode(y, dt))


@pytest.mark.parametrize("shape", [(1, 8, 32, 32), (2, 5, 18, 22), (1, 4, 17, 19)])
@pytest.mark.parametrize("dt", [_capi.F32, _capi.BF16])
def test_stem(shape, dt):
    N, T, H, W = shape
    rs = np.random.RandomState(T + H)
    x = rs.uniform(-1, 1, (N, T, 3, H, W)).astype(np.float32)
    Cout = 64 if H == 32 else 40
    w = (rs.randn(Cout, 3, 7, 7, 7) / np.sqrt(1029)).astype(np.float32)
    scale = (1 + 0.1 * rs.randn(Cout)).astype(np.float32)
    shift = (0.2 * rs.randn(Cout)).astype(np.float32)
    got = run_stem(x, w, scale, shift, dt)
    xq = torch.from_numpy(E.quantize(x, dt)).permute(0, 2, 1, 3, 4)
    y = F.conv3d(F.pad(xq, (2, 3, 2, 3, 2, 3)), torch.from_numpy(E.quantize(w, dt)), stride=2)
    ref = F.relu(y * torch.from_numpy(scale).view(1, -1, 1, 1, 1) + torch.from_numpy(shift).view(1, -1, 1, 1, 1)).numpy()
    assert got.shape == ref.shape
    err = np.abs(got - ref).max() / np.abs(ref).max()
    assert err < tol(dt), err
