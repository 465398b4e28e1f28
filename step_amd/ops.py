"""step_amd/ops.py -- torch-tensor front-end to the C ABI (include/step_amd.h).

PyTorch is used here only for device memory, streams and autograd bookkeeping; every operator
below is one call into libstep_amd.so.  Activations are CHANNELS-LAST: a 5-D activation is a
contiguous tensor [N, D, H, W, C] (possibly a channel slice [.., c0:c1] of a wider buffer).
"""
import ctypes
import os

import torch

from . import _capi, _lib

DT = {torch.float32: _capi.F32, torch.bfloat16: _capi.BF16, torch.float16: _capi.F16}

# Optional per-launch instrumentation (bench.py's roofline legs).  PROFILE = a list: every instrumented launch appends
# (kernel name, algorithmic FLOPs, algorithmic bytes, event0, event1).
#   * PROFILE_LIMIT = None (eager legs of the C3 / C4 pipelines): HIP events bracket the launch on its launch stream;
#   * PROFILE_LIMIT = k (the C2 leg): nothing is timed here -- the launch is recorded and EXECUTED only if it is among the first k of
#     the forward, so that bench.py can capture HIP graphs of growing prefixes of the step and time a launch as the difference of two
#     replays: in its real place of the replayed sequence, at replay clocks, its inputs wherever the previous kernel left them.
PROFILE = None
PROFILE_LIMIT = None
WGRAD_WS = True        # fp32 weight gradients: partial-tile workspace + fixed-order sum instead of fp32 atomics (bit-reproducible; module switch for tests / A-B timing)
WGRAD16_WS = True      # 16-bit weight gradients: partial-tile workspace + fixed-order sum instead of fp32 atomics (module switch for tests / A-B timing)
class _Prof:
    __slots__ = ("name", "flops", "bytes", "e0")

    def __init__(self, name, flops, nbytes):
        self.name, self.flops, self.bytes = name, flops, nbytes

    def __enter__(self):
        self.e0 = torch.cuda.Event(enable_timing=True)
        self.e0.record()
        return self

    def __exit__(self, *exc):
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        PROFILE.append((self.name, self.flops, self.bytes, self.e0, e1))
        return False


class _NoProf:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NOPROF = _NoProf()


def _run(launch, describe):
    """Run one forward launch; under PROFILE record / time it (see the comment at PROFILE).  describe() -> (name, flops, bytes)."""
    if PROFILE is None:
        launch()
        return
    name, flops, nbytes = describe()
    if PROFILE_LIMIT is not None:
        PROFILE.append((name, flops, nbytes, None, None))
        if len(PROFILE) <= PROFILE_LIMIT:
            launch()
        return
    with _Prof(name, flops, nbytes):
        launch()
_ES = {torch.float32: 4, torch.bfloat16: 2, torch.float16: 2}
_TNAME = {torch.float32: "float", torch.bfloat16: "step::bf16_t", torch.float16: "step::f16_t"}


def _dt(t):
    try:
        return DT[t.dtype]
    except KeyError:
        raise RuntimeError("step_amd: unsupported dtype %s (float32 / bfloat16 / float16)" % t.dtype)


def _chan_slice(t):
    """Channel stride (elements between consecutive pixels) of a channels-last tensor that may be a slice
    of the last dim of a wider dense buffer; raises if the tensor is anything else."""
    if t.stride(-1) != 1:
        raise RuntimeError("step_amd: activation is not channels-last")
    cs = t.stride(-2) if t.dim() >= 2 else t.size(-1)
    # all leading dims must be dense over a [.., cs] buffer
    exp = cs
    for i in range(t.dim() - 2, -1, -1):
        if t.size(i) != 1 and t.stride(i) != exp:
            raise RuntimeError("step_amd: activation must be a channel slice of a dense channels-last buffer")
        exp *= t.size(i)
    return cs


def conv_packed_elems(Cout, Cin, k):
    return _lib.lib().step_conv_packed_elems(Cout, Cin, k[0], k[1], k[2])


def pack_conv_weight(w, dtype, perm=None):
    """w: fp32 [Cout, Cin, kd, kh, kw] (or 2-D/4-D, reshaped by the caller) on the device."""
    L = _lib.lib()
    w = w.detach().contiguous().float()
    Cout, Cin, kd, kh, kw = w.shape
    out = torch.empty(L.step_conv_packed_elems(Cout, Cin, kd, kh, kw), dtype=dtype, device=w.device)
    _capi.check(L.step_conv_pack_weight(_lib.dptr(w), Cout, Cin, kd, kh, kw, DT[dtype], _lib.dptr(perm), _lib.dptr(out),
                                        _lib.stream_ptr(w.device)), "step_conv_pack_weight")
    return out


def pack_conv_weight_dgrad(w, dtype, cin_pad):
    """Packed weight of the data-gradient conv (taps flipped, channel roles swapped, input channels padded to cin_pad) from
    the forward weight w [Cout, Cin, kd, kh, kw] in one kernel."""
    L = _lib.lib()
    w = w.detach().contiguous().float()
    Cout, Cin, kd, kh, kw = w.shape
    out = torch.empty(L.step_conv_packed_elems(Cin, cin_pad, kd, kh, kw), dtype=dtype, device=w.device)
    _capi.check(L.step_conv_pack_weight_dgrad(_lib.dptr(w), Cout, Cin, kd, kh, kw, DT[dtype], cin_pad, _lib.dptr(out),
                                              _lib.stream_ptr(w.device)), "step_conv_pack_weight_dgrad")
    return out


def pack_table(entries, device):
    """Device array of step_pack_item for pack_conv_weights.  entries: (w_ptr, perm_ptr | None, packed_ptr, Cout, Cin, w_cin,
    cin_lo, (kd, kh, kw), dgrad, cin_pad) per weight; Cout / Cin are the FORWARD conv's (effective) channel counts."""
    items = (_capi.PackItem * len(entries))()
    for it, (w, perm, dst, cout, cin, wcin, lo, k, dgrad, cpad) in zip(items, entries):
        if not (cout > 0 and cin > 0 and 0 <= lo and lo + cin <= wcin and min(k) > 0) or (dgrad and (cpad < cout or not all(v & 1 for v in k))):
            raise ValueError("pack_table: bad item %r" % ((cout, cin, wcin, lo, k, dgrad, cpad),))
        it.w, it.perm_c, it.packed = w, perm, dst
        it.Cout, it.Cin, it.w_cin, it.cin_lo = cout, cin, wcin, lo
        it.kd, it.kh, it.kw = k
        it.dgrad, it.cin_pad = int(bool(dgrad)), cpad
    host = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8)
    return host.to(device)


def pack_conv_weights(table, n, dtype, device):
    """Every weight of a net re-packed by one launch (step_conv_pack_weights); `table` from pack_table()."""
    L = _lib.lib()
    _capi.check(L.step_conv_pack_weights(_lib.dptr(table), n, DT[dtype], _lib.stream_ptr(device)), "step_conv_pack_weights")


def pack_stem_weight(w, dtype):
    L = _lib.lib()
    w = w.detach().contiguous().float()
    out = torch.empty(L.step_stem_packed_elems(w.shape[0]), dtype=dtype, device=w.device)
    _capi.check(L.step_stem_pack_weight(_lib.dptr(w), w.shape[0], DT[dtype], _lib.dptr(out), _lib.stream_ptr(w.device)),
                "step_stem_pack_weight")
    return out


def conv_forward(x, w_packed, Cout, k, scale=None, shift=None, relu=True, res=None, out=None, out2=None, split=0):
    """x: channels-last [N,D,H,W,Cin] (may be a channel slice); out: optional channel slice to write
    into (same N,D,H,W, Cout channels); returns out.  With split > 0 (1x1x1 only) output channels
    [0, split) go to `out` and [split, Cout) to `out2`."""
    L = _lib.lib()
    N, D, H, W, Cin = x.shape
    xcs = _chan_slice(x)
    if out is None:
        out = torch.empty((N, D, H, W, split if split else Cout), dtype=x.dtype, device=x.device)
    ycs = _chan_slice(out)
    d = _capi.ConvDesc(dtype=_dt(x), N=N, D=D, H=H, W=W, Cin=Cin, Cout=Cout, kd=k[0], kh=k[1], kw=k[2],
                       x_cstride=xcs, x_coff=0, y_cstride=ycs, y_coff=0,
                       res_cstride=(_chan_slice(res) if res is not None else 0), res_coff=0, relu=int(bool(relu)),
                       split=int(split), y2_cstride=(_chan_slice(out2) if out2 is not None else 0), y2_coff=0)
    def describe():
        buf = ctypes.create_string_buffer(256)
        L.step_conv_kernel_name(ctypes.byref(d), buf, 256)
        pix = N * D * H * W
        return (buf.value.decode(), 2.0 * pix * Cout * Cin * k[0] * k[1] * k[2],
                (pix * (Cin + Cout) + Cout * Cin * k[0] * k[1] * k[2]) * _ES[x.dtype])
    wsb = L.step_conv_workspace_bytes(ctypes.byref(d))            # > 0 only for the split-K Linear layers of the heads
    ws = torch.empty(wsb, dtype=torch.uint8, device=x.device) if wsb else None
    def launch():
        _capi.check(L.step_conv_forward_ws(ctypes.byref(d), _lib.dptr(x), _lib.dptr(w_packed), _lib.dptr(scale), _lib.dptr(shift),
                                           _lib.dptr(res), _lib.dptr(out), _lib.dptr(out2), _lib.dptr(ws), wsb,
                                           _lib.stream_ptr(x.device)), "step_conv_forward_ws")
    _run(launch, describe)
    return out


def conv_forward_cat(xa, xb, w_packed, Cout, scale=None, shift=None, relu=True, res=None, out=None):
    """Pointwise conv over the channel concat [xa | xb] that is never materialised (step_conv_forward_cat: conv1 / conv2 of the heads'
    resample Bottleneck, two_branch.py:86-111): ONE launch and one fp32 accumulation over the whole K instead of two accumulating
    launches with the partial sum rounded to the storage type in between.  w_packed: the packed image of the WHOLE weight
    [Cout, Ca + Cb].  Returns None when the library has no such form for the shapes (the caller launches the halves)."""
    L = _lib.lib()
    N, D, H, W, Ca = xa.shape
    Cb = xb.shape[-1]
    if xa.dtype == torch.float32 or xb.dtype != xa.dtype or tuple(xb.shape[:4]) != (N, D, H, W) or Ca % 32 or Cb % 8:
        return None
    if out is None:
        out = torch.empty((N, D, H, W, Cout), dtype=xa.dtype, device=xa.device)
    d = _capi.ConvDesc(dtype=_dt(xa), N=N, D=D, H=H, W=W, Cin=Ca + Cb, Cout=Cout, kd=1, kh=1, kw=1, x_cstride=_chan_slice(xa), x_coff=0,
                       y_cstride=_chan_slice(out), y_coff=0, res_cstride=(_chan_slice(res) if res is not None else 0), res_coff=0,
                       relu=int(bool(relu)), split=0, y2_cstride=0, y2_coff=0)
    dp = _capi.ConvDesc(dtype=_dt(xa), N=N, D=D, H=H, W=W, Cin=Ca + Cb, Cout=Cout, kd=1, kh=1, kw=1, x_cstride=Ca + Cb, x_coff=0,
                        y_cstride=_chan_slice(out), y_coff=0, res_cstride=0, res_coff=0, relu=int(bool(relu)), split=0, y2_cstride=0, y2_coff=0)
    info = (ctypes.c_int * 10)()
    if L.step_conv_plan_info(ctypes.byref(dp), info, 10) != 0 or info[0] != 2:
        return None                                                      # (the streaming GEMM only: the test step_conv_forward_cat makes)
    refused = []

    def launch():
        rc = L.step_conv_forward_cat(ctypes.byref(d), _lib.dptr(xa), int(Ca), _lib.dptr(xb), _chan_slice(xb), 0, _lib.dptr(w_packed),
                                     _lib.dptr(scale), _lib.dptr(shift), _lib.dptr(res), _lib.dptr(out), None, _lib.stream_ptr(xa.device))
        if rc in (-4, -5):
            refused.append(rc)
            return
        _capi.check(rc, "step_conv_forward_cat")

    def describe():
        pix = N * D * H * W
        return ("void step::conv_pw2_kernel<%s, %d, %d>(step::ConvParams)" % (_TNAME[xa.dtype], min(info[2], 2) if info[3] == 8 else info[2], info[3]),
                2.0 * pix * Cout * (Ca + Cb), (pix * (Ca + Cb + Cout * (2 if res is not None else 1)) + Cout * (Ca + Cb)) * _ES[xa.dtype])
    _run(launch, describe)
    if refused:
        if PROFILE is not None and PROFILE and PROFILE[-1][0].startswith("void step::conv_pw2_kernel"):
            PROFILE.pop()
        return None
    return out


POOL_CONV_FORCE_NB = 0  # pool_conv_forward: 1 | 2 force the accumulator depth of the pointwise workgroups inside the combined grid (A/B timing; 0: the library's choice)
POOL_CONV_MAX_NB = 2   # pool_conv_forward: deepest accumulator form of the standalone pointwise launch that still rides with the pool (module switch for A/B timing)


def pool_conv_forward(x, w_packed, Cout, scale, shift, relu, out, out2=None, split=0):
    """The 3x3x3 / 1 TF-SAME max pool of x AND a pointwise conv of x as one launch (step_pool_conv_forward: an Inception block's
    branch_3 pool beside its fused 1x1x1 triple).  Returns the pooled tensor, or None when the library keeps the two apart for these
    shapes (the caller then launches them one after the other; same results bit for bit)."""
    L = _lib.lib()
    N, D, H, W, Cin = x.shape
    if x.dtype == torch.float32:
        return None
    d = _capi.ConvDesc(dtype=_dt(x), N=N, D=D, H=H, W=W, Cin=Cin, Cout=Cout, kd=1, kh=1, kw=1, x_cstride=_chan_slice(x), x_coff=0,
                       y_cstride=_chan_slice(out), y_coff=0, res_cstride=0, res_coff=0, relu=int(bool(relu)), split=int(split),
                       y2_cstride=(_chan_slice(out2) if out2 is not None else 0), y2_coff=0)
    info = (ctypes.c_int * 10)()
    if L.step_conv_plan_info(ctypes.byref(d), info, 10) != 0 or info[0] != 2 or info[2] > POOL_CONV_MAX_NB:
        return None                                                      # (the test step_pool_conv_forward makes)
    force = {"conv_nb": POOL_CONV_FORCE_NB} if POOL_CONV_FORCE_NB else {}
    with _capi.options(L, **force):
        nbc = L.step_pool_conv_plan_nb(ctypes.byref(d))
    if nbc <= 0:
        return None
    pooled = torch.empty((N, D, H, W, Cin), dtype=x.dtype, device=x.device)
    refused = []

    def launch():
        with _capi.options(L, **force):
            rc = L.step_pool_conv_forward(_dt(x), _lib.dptr(x), N, D, H, W, Cin, _chan_slice(x), 0, _lib.dptr(pooled), Cin, 0, ctypes.byref(d),
                                          _lib.dptr(x), _lib.dptr(w_packed), _lib.dptr(scale), _lib.dptr(shift), _lib.dptr(out), _lib.dptr(out2),
                                          _lib.stream_ptr(x.device))
        if rc in (-4, -5):          # STEP_E_UNSUPPORTED / STEP_E_ALIGN: the library's own contract checks (alignment, 32-bit offsets, ...) are
            refused.append(rc)      # stricter than the plan test above -- the caller then launches pool and conv one after the other
            return
        _capi.check(rc, "step_pool_conv_forward")

    def describe():
        pix = N * D * H * W
        return ("void step::pool333_pw_kernel<%s, %d>(%s const*, %s*, step::PoolParams, int, int, int, int, int, int, int, int, step::ConvParams)" % (
            _TNAME[x.dtype], nbc, _TNAME[x.dtype], _TNAME[x.dtype]), 2.0 * pix * Cout * Cin, (pix * (3 * Cin + Cout) + Cout * Cin) * _ES[x.dtype])
    _run(launch, describe)
    if refused:
        if PROFILE is not None and PROFILE and PROFILE[-1][0].startswith("void step::pool333_pw_kernel"):
            PROFILE.pop()
        return None
    return pooled


def conv_forward_pre(x, w_packed, Cout, k, scale, shift, relu, pre, out=None):
    """conv_forward whose input is a pointwise conv + affine + ReLU of x, evaluated on the fly (step_conv_forward_pre: conv3d_2b in
    front of conv3d_2c; the tensor between them never exists).  pre = (packed pointwise weight, scale, shift, Cmid): x has the
    pointwise layer's input channels, Cmid = its output channels = this conv's input channels.  Returns None when the library has
    no fused form for the layer (the caller launches the two layers one after the other); bit-identical to that."""
    L = _lib.lib()
    pw, pscale, pshift, cmid = pre
    N, D, H, W, Cpre = x.shape
    if out is None:
        out = torch.empty((N, D, H, W, Cout), dtype=x.dtype, device=x.device)
    d = _capi.ConvDesc(dtype=_dt(x), N=N, D=D, H=H, W=W, Cin=cmid, Cout=Cout, kd=k[0], kh=k[1], kw=k[2],
                       x_cstride=_chan_slice(x), x_coff=0, y_cstride=_chan_slice(out), y_coff=0, res_cstride=0, res_coff=0,
                       relu=int(bool(relu)), split=0, y2_cstride=0, y2_coff=0)
    info = (ctypes.c_int * 10)()
    if Cpre != 64 or cmid != 64 or x.dtype == torch.float32 or tuple(k) != (3, 3, 3) or L.step_conv_plan_info(ctypes.byref(d), info, 10) != 0 \
            or not (info[0] == 1 and info[4] == 1 and info[3] == 8 and info[1] in (0, 3)):
        return None                                                      # (the same test step_conv_forward_pre makes: STEP_E_UNSUPPORTED)

    refused = []

    def launch():
        rc = L.step_conv_forward_pre(ctypes.byref(d), _lib.dptr(x), _lib.dptr(w_packed), _lib.dptr(scale), _lib.dptr(shift), _lib.dptr(pw),
                                     _lib.dptr(pscale), _lib.dptr(pshift), int(Cpre), _lib.dptr(out), _lib.stream_ptr(x.device))
        if rc in (-4, -5):          # the library's stricter contract (alignment of x / scale tables, x_coff, 32-bit offsets): fall back to two launches
            refused.append(rc)
            return
        _capi.check(rc, "step_conv_forward_pre")

    def describe():
        pix = N * D * H * W
        return ("void step::conv_tap_pre_kernel<%s, %d, %d>(step::ConvParams)" % (_TNAME[x.dtype], info[1], info[2]),
                2.0 * pix * (Cout * cmid * 27 + cmid * Cpre), (pix * (Cpre + Cout) + Cout * cmid * 27 + cmid * Cpre) * _ES[x.dtype])
    _run(launch, describe)
    if refused:
        if PROFILE is not None and PROFILE and PROFILE[-1][0].startswith("void step::conv_tap_pre_kernel"):
            PROFILE.pop()
        return None
    return out


def conv_forward_pre_pool(x, w_packed, Cout, k, scale, shift, relu, pre):
    """conv_forward_pre with the (1,3,3) / (1,2,2) max pool behind it taken on the conv's tiles (step_conv_forward_pre_pool: conv3d_2b ->
    conv3d_2c -> maxPool3d_3a as one call; neither the tensor between the convs nor the un-pooled conv output exists).  Returns the
    POOLED channels-last tensor [N,D,Hp,Wp,Cout], or None when the library has no fused form for the layer (the caller keeps
    conv_forward_pre + the pool; bit-identical)."""
    L = _lib.lib()
    pw, pscale, pshift, cmid = pre
    N, D, H, W, Cpre = x.shape
    if Cpre != 64 or cmid != 64 or x.dtype == torch.float32 or tuple(k) != (3, 3, 3) or not relu:
        return None
    Hp, Wp = L.step_pool_out_size(H, 3, 2), L.step_pool_out_size(W, 3, 2)
    d = _capi.ConvDesc(dtype=_dt(x), N=N, D=D, H=H, W=W, Cin=cmid, Cout=Cout, kd=3, kh=3, kw=3,
                       x_cstride=_chan_slice(x), x_coff=0, y_cstride=Cout, y_coff=0, res_cstride=0, res_coff=0,
                       relu=1, split=0, y2_cstride=0, y2_coff=0)
    wsb = L.step_conv_pre_pool_workspace_bytes(ctypes.byref(d))
    if not wsb:
        return None
    out = torch.empty((N, D, Hp, Wp, Cout), dtype=x.dtype, device=x.device)
    ws = torch.empty(wsb, dtype=torch.uint8, device=x.device)
    info = (ctypes.c_int * 13)()
    if L.step_conv_pre_pool_plan_info(ctypes.byref(d), info, 13) != 0:     # the plan the POOLED call uses (it may re-plan a general-box layer onto 4 x 8 x 8 tiles)
        return None
    refused = []

    def launch():
        # (the call in its two parts so that the instrumented legs time the conv launches and the seam pass separately)
        rc = L.step_conv_forward_pre_pool_tiles(ctypes.byref(d), _lib.dptr(x), _lib.dptr(w_packed), _lib.dptr(scale), _lib.dptr(shift), _lib.dptr(pw),
                                                _lib.dptr(pscale), _lib.dptr(pshift), int(Cpre), _lib.dptr(out), _lib.dptr(ws), wsb, _lib.stream_ptr(x.device))
        if rc in (-4, -5):          # the library's stricter contract (alignment, 32-bit offsets): the caller falls back
            refused.append(rc)
            return
        _capi.check(rc, "step_conv_forward_pre_pool_tiles")

    def finish():
        _capi.check(L.step_conv_pre_pool_finish(ctypes.byref(d), _lib.dptr(out), _lib.dptr(ws), wsb, _lib.stream_ptr(x.device)), "step_conv_pre_pool_finish")

    def describe_finish():
        th, tw = info[10], info[11]
        seam = N * D * ((th - 1) * Wp + (tw - 1) * Hp) * Cout
        rows = N * D * (th * W + tw * H) * Cout
        return ("step::pool_seam_fix_kernel(step::PoolFixParams)", 0.0, (2 * seam + rows) * _ES[x.dtype])

    def describe():
        pix = N * D * H * W
        return ("void step::conv_tap_pre_pool%s_kernel<%s, %d>(step::ConvParams)" % ("_persist" if info[12] else "", _TNAME[x.dtype], info[2]),
                2.0 * pix * (Cout * cmid * 27 + cmid * Cpre), (pix * Cpre + N * D * Hp * Wp * Cout + Cout * cmid * 27 + cmid * Cpre) * _ES[x.dtype])
    _run(launch, describe)
    if refused:
        if PROFILE is not None and PROFILE and PROFILE[-1][0].startswith("void step::conv_tap_pre_pool"):
            PROFILE.pop()
        return None
    _run(finish, describe_finish)
    return out


def conv_forward_group(members):
    """Several INDEPENDENT convs as one launch where the library can merge them (step_conv_forward_group: today two 16-bit 3x3x3
    layers of the two-phase conv_tap form -- an Inception block's branch_1 / branch_2 convs -- plus, on small maps, one pointwise
    conv riding on the CUs they leave idle); otherwise they are launched one after the other.  members: (x, w_packed, Cout, k, scale, shift, relu, out) per conv, `out` a channel slice to write into.  Same results
    as conv_forward per member, bit for bit."""
    L = _lib.lib()
    # a member whose plan needs a caller-owned workspace (the split-K form of few-row / deep-K pointwise layers) goes through
    # conv_forward, which allocates it: inside the group call it would silently run on the tiled kernel instead -- a valid result, but
    # summed in another order than the same layer launched alone (bit-identity of the two forms is part of this function's contract)
    solo = []
    for m_ in members:
        x, w_packed, Cout, k, scale, shift, relu, out = m_
        if tuple(k) == (1, 1, 1):
            N, D, H, W, Cin = x.shape
            d = _capi.ConvDesc(dtype=_dt(x), N=N, D=D, H=H, W=W, Cin=Cin, Cout=Cout, kd=1, kh=1, kw=1, x_cstride=_chan_slice(x), x_coff=0,
                               y_cstride=_chan_slice(out), y_coff=0, res_cstride=0, res_coff=0, relu=int(bool(relu)), split=0, y2_cstride=0, y2_coff=0)
            if L.step_conv_workspace_bytes(ctypes.byref(d)):
                solo.append(m_)
    if solo:
        members = [m_ for m_ in members if not any(m_ is s_ for s_ in solo)]
        for (x, w_packed, Cout, k, scale, shift, relu, out) in solo:
            conv_forward(x, w_packed, Cout, k, scale, shift, relu, None, out)
        if not members:
            return
        if len(members) == 1:
            x, w_packed, Cout, k, scale, shift, relu, out = members[0]
            conv_forward(x, w_packed, Cout, k, scale, shift, relu, None, out)
            return
    n = len(members)
    items = (_capi.ConvItem * n)()
    descs, flops, nbytes = [], 0.0, 0
    for it, (x, w_packed, Cout, k, scale, shift, relu, out) in zip(items, members):
        N, D, H, W, Cin = x.shape
        d = _capi.ConvDesc(dtype=_dt(x), N=N, D=D, H=H, W=W, Cin=Cin, Cout=Cout, kd=k[0], kh=k[1], kw=k[2],
                           x_cstride=_chan_slice(x), x_coff=0, y_cstride=_chan_slice(out), y_coff=0, res_cstride=0, res_coff=0,
                           relu=int(bool(relu)), split=0, y2_cstride=0, y2_coff=0)
        descs.append(d)
        it.desc = ctypes.pointer(d)
        it.x, it.w_packed, it.y = x.data_ptr(), w_packed.data_ptr(), out.data_ptr()
        it.scale = None if scale is None else scale.data_ptr()
        it.shift = None if shift is None else shift.data_ptr()
        it.res = None
        _lib.dptr(x), _lib.dptr(out)                                # (refuse non-device tensors: there is no CPU fallback)
        pix = N * D * H * W
        flops += 2.0 * pix * Cout * Cin * k[0] * k[1] * k[2]
        nbytes += (pix * (Cin + Cout) + Cout * Cin * k[0] * k[1] * k[2]) * _ES[x.dtype]
    stream = _lib.stream_ptr(members[0][0].device)
    if PROFILE is not None:
        buf = ctypes.create_string_buffer(256)
        L.step_conv_group_kernel_name(items, n, buf, 256)
        if not buf.value:                                            # not merged: per-member attribution
            for (x, w_packed, Cout, k, scale, shift, relu, out) in members:
                conv_forward(x, w_packed, Cout, k, scale, shift, relu, None, out)
            return
        pw = [m for m in members if tuple(m[3]) == (1, 1, 1)]
        if pw and b"_pw_kernel" not in buf.value:                    # the pointwise member stays a launch of its own: attribute it separately
            conv_forward_group([m for m in members if tuple(m[3]) != (1, 1, 1)])
            for (x, w_packed, Cout, k, scale, shift, relu, out) in pw:
                conv_forward(x, w_packed, Cout, k, scale, shift, relu, None, out)
            return
    _run(lambda: _capi.check(L.step_conv_forward_group(items, n, stream), "step_conv_forward_group"),
         lambda: (buf.value.decode(), flops, nbytes))


def _grad_slice(gy, vec):
    """Channel stride of a gradient that is a channel slice of a dense channels-last buffer the weight-gradient kernels can read in place
    (stride and start on the kernels' vector grid), else None (the caller densifies)."""
    try:
        cs = _chan_slice(gy)
    except RuntimeError:
        return None
    if cs % vec or gy.data_ptr() % 16:
        return None
    return cs


def conv_wgrad(x, gy, Cout, k, into=None):
    """Weight gradient of the stride-1 SAME conv: x channels-last [N,D,H,W,Cin] (any storage dtype, may be a channel
    slice), gy fp32 channels-last [N,D,H,W,Cout] (gradient w.r.t. the conv output before the affine epilogue).
    Returns fp32 [Cout, Cin, kd, kh, kw]; with `into` (a dense fp32 tensor of that many elements, e.g. the parameter's
    .grad) the result is ACCUMULATED into it instead (the kernel's accumulate mode: no clear, no separate add)."""
    L = _lib.lib()
    N, D, H, W, Cin = x.shape
    if gy.dtype != torch.float32:
        gy = gy.float()
    gcs = _grad_slice(gy, 4)
    if gcs is None:
        gy, gcs = gy.contiguous(), Cout
    if into is not None:
        if into.dtype != torch.float32 or not into.is_contiguous() or into.numel() != Cout * Cin * k[0] * k[1] * k[2]:
            raise RuntimeError("step_amd: conv_wgrad(into=...) wants a dense fp32 tensor of Cout*Cin*taps elements")
        dw = into
    else:
        dw = torch.empty((Cout, Cin) + tuple(k), dtype=torch.float32, device=x.device)
    d = _capi.ConvDesc(dtype=_dt(x), N=N, D=D, H=H, W=W, Cin=Cin, Cout=Cout, kd=k[0], kh=k[1], kw=k[2],
                       x_cstride=_chan_slice(x), x_coff=0, y_cstride=gcs, y_coff=0, res_cstride=0, res_coff=0, relu=0,
                       split=0, y2_cstride=0, y2_coff=0)
    prof = _NOPROF
    if PROFILE is not None:
        pix = N * D * H * W
        # algorithmic: x and dy read once, dw written once (fp32); the kernel variant is chosen by Cin inside the library
        nbuf = ctypes.create_string_buffer(256)
        L.step_conv_wgrad_kernel_name(ctypes.byref(d), 0, nbuf, 256)
        prof = _Prof(nbuf.value.decode(), 2.0 * pix * Cout * Cin * k[0] * k[1] * k[2],
                     pix * (Cin * x.element_size() + Cout * 4) + 4.0 * Cout * Cin * k[0] * k[1] * k[2])
    wsb = L.step_conv_wgrad_workspace_bytes(ctypes.byref(d)) if WGRAD_WS else 0     # partial tiles + fixed-order sum: no atomics, deterministic
    ws = torch.empty(wsb, dtype=torch.uint8, device=x.device) if wsb else None
    with prof:
        _capi.check(L.step_conv_wgrad_ws(ctypes.byref(d), _lib.dptr(x), _lib.dptr(gy), _lib.dptr(dw), int(into is not None),
                                         _lib.dptr(ws), wsb, _lib.stream_ptr(x.device)), "step_conv_wgrad_ws")
    return dw


def conv_wgrad16(x, gy, Cout, k, into=None):
    """conv_wgrad on the 16-bit matrix instructions: x and gy both in the 16-bit activation dtype (gy dense channels-last
    [N,D,H,W,Cout]); fp32 result / accumulation as conv_wgrad."""
    L = _lib.lib()
    N, D, H, W, Cin = x.shape
    if gy.dtype != x.dtype or x.dtype == torch.float32:
        raise RuntimeError("step_amd: conv_wgrad16 wants x and gy in the same 16-bit dtype")
    gcs = _grad_slice(gy, 8)
    if gcs is None:
        gy, gcs = gy.contiguous(), Cout
    if into is not None:
        if into.dtype != torch.float32 or not into.is_contiguous() or into.numel() != Cout * Cin * k[0] * k[1] * k[2]:
            raise RuntimeError("step_amd: conv_wgrad16(into=...) wants a dense fp32 tensor of Cout*Cin*taps elements")
        dw = into
    else:
        dw = torch.empty((Cout, Cin) + tuple(k), dtype=torch.float32, device=x.device)
    d = _capi.ConvDesc(dtype=_dt(x), N=N, D=D, H=H, W=W, Cin=Cin, Cout=Cout, kd=k[0], kh=k[1], kw=k[2],
                       x_cstride=_chan_slice(x), x_coff=0, y_cstride=gcs, y_coff=0, res_cstride=0, res_coff=0, relu=0,
                       split=0, y2_cstride=0, y2_coff=0)
    prof = _NOPROF
    if PROFILE is not None:
        pix = N * D * H * W
        nbuf = ctypes.create_string_buffer(256)
        L.step_conv_wgrad_kernel_name(ctypes.byref(d), 1, nbuf, 256)
        prof = _Prof(nbuf.value.decode(), 2.0 * pix * Cout * Cin * k[0] * k[1] * k[2],
                     pix * (Cin + Cout) * x.element_size() + 4.0 * Cout * Cin * k[0] * k[1] * k[2])
    wsb = L.step_conv_wgrad16_workspace_bytes(ctypes.byref(d)) if WGRAD16_WS else 0   # > 0: the LDS-tiled form with a two-stage sum
    ws = torch.empty(wsb, dtype=torch.uint8, device=x.device) if wsb else None
    with prof:
        _capi.check(L.step_conv_wgrad16_ws(ctypes.byref(d), _lib.dptr(x), _lib.dptr(gy), _lib.dptr(dw), int(into is not None),
                                           _lib.dptr(ws), wsb, _lib.stream_ptr(x.device)), "step_conv_wgrad16_ws")
    return dw


def conv_wgrad_partial(x, gy, Cout, k, into=None):
    """conv_wgrad / conv_wgrad16 (chosen by gy's dtype) with the fixed-order sum of the partial tiles DEFERRED: launches the kernel that
    writes the partial tiles and returns (dw, item, ws) -- pass the items of several layers to wgrad_reduce_group (one launch) and keep
    `ws` alive until then.  item.kind == 0: nothing is pending."""
    L = _lib.lib()
    N, D, H, W, Cin = x.shape
    w16 = gy.dtype == x.dtype and x.dtype != torch.float32
    if not w16 and gy.dtype != torch.float32:
        gy = gy.float()
    gcs = _grad_slice(gy, 8 if w16 else 4)
    if gcs is None:
        gy, gcs = gy.contiguous(), Cout
    if into is not None:
        if into.dtype != torch.float32 or not into.is_contiguous() or into.numel() != Cout * Cin * k[0] * k[1] * k[2]:
            raise RuntimeError("step_amd: conv_wgrad_partial(into=...) wants a dense fp32 tensor of Cout*Cin*taps elements")
        dw = into
    else:
        dw = torch.empty((Cout, Cin) + tuple(k), dtype=torch.float32, device=x.device)
    d = _capi.ConvDesc(dtype=_dt(x), N=N, D=D, H=H, W=W, Cin=Cin, Cout=Cout, kd=k[0], kh=k[1], kw=k[2],
                       x_cstride=_chan_slice(x), x_coff=0, y_cstride=gcs, y_coff=0, res_cstride=0, res_coff=0, relu=0,
                       split=0, y2_cstride=0, y2_coff=0)
    wsb = (L.step_conv_wgrad16_workspace_bytes if w16 else L.step_conv_wgrad_workspace_bytes)(ctypes.byref(d))
    ws = torch.empty(wsb, dtype=torch.uint8, device=x.device) if wsb else None
    item = _capi.WgradReduceItem()
    _capi.check(L.step_conv_wgrad_partial(ctypes.byref(d), _lib.dptr(x), _lib.dptr(gy), int(w16), _lib.dptr(dw), int(into is not None), _lib.dptr(ws), wsb,
                                          ctypes.byref(item), _lib.stream_ptr(x.device)), "step_conv_wgrad_partial")
    return dw, item, ws


def wgrad_reduce_group(items, device):
    """The deferred sums of conv_wgrad_partial, up to 8 per launch (step_wgrad_reduce_group)."""
    L = _lib.lib()
    live = [it for it in items if it.kind != 0]
    for i in range(0, len(live), _capi.WGRAD_REDUCE_MAX):
        chunk = live[i:i + _capi.WGRAD_REDUCE_MAX]
        arr = (_capi.WgradReduceItem * len(chunk))(*chunk)
        _capi.check(L.step_wgrad_reduce_group(arr, len(chunk), _lib.stream_ptr(device)), "step_wgrad_reduce_group")


def act_grad(y, gy, scale, relu, want_f32=True, want_act=False):
    """Activation gradient of the fused conv unit: g = gy * (y > 0) * scale[c] in one pass (step_act_grad).
    y / gy channels-last (possibly channel slices); returns (g fp32 or None, g in y.dtype or None), dense.
    Returns None when the layout is outside the kernel's contract (channels not a multiple of 4): the caller then keeps
    torch's element-wise ops."""
    C = y.shape[-1]
    if C % 4 or gy.dtype not in (torch.float32, y.dtype) or gy.stride(-1) != 1 or y.stride(-1) != 1:
        return None
    try:
        ycs, gcs = _chan_slice(y), _chan_slice(gy)
    except RuntimeError:
        return None
    es_y, es_g = y.element_size(), gy.element_size()
    if ycs % 4 or gcs % 4 or y.data_ptr() % (4 * es_y) or gy.data_ptr() % (4 * es_g):
        return None
    M = y.numel() // C
    same = y.dtype == torch.float32
    g32 = torch.empty(y.shape, dtype=torch.float32, device=y.device) if (want_f32 or (want_act and same)) else None
    gact = torch.empty(y.shape, dtype=y.dtype, device=y.device) if (want_act and not same) else None
    sc = None
    if scale is not None:
        sc = scale if (scale.dtype == torch.float32 and scale.is_contiguous()) else scale.float().contiguous()
    L = _lib.lib()
    _capi.check(L.step_act_grad(_dt(y), _lib.dptr(y), ycs, DT[gy.dtype], _lib.dptr(gy), gcs, _lib.dptr(sc), M, C, int(bool(relu)),
                                _lib.dptr(g32), _lib.dptr(gact), _lib.stream_ptr(y.device)), "step_act_grad")
    return (g32 if want_f32 or same else None), (g32 if same else gact)


def bn_train_forward(z, gamma, beta, running_mean, running_var, eps, momentum, relu, out=None):
    """Batch-statistics BatchNorm (+ ReLU) of a raw conv output z (channels-last, possibly a channel slice): step_bn_train_forward.
    Updates running_mean / running_var in place (None: no update).  Returns (y, save_mean, save_invstd)."""
    L = _lib.lib()
    C = z.shape[-1]
    M = z.numel() // C
    zcs = _chan_slice(z)
    if out is None:
        out = torch.empty(z.shape, dtype=z.dtype, device=z.device)
    f32 = lambda t: None if t is None else (t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous())
    g_, b_ = f32(gamma), f32(beta)
    for t in (running_mean, running_var):
        if t is not None and (t.dtype != torch.float32 or not t.is_contiguous()):
            raise RuntimeError("step_amd: BatchNorm running statistics must be contiguous fp32 tensors")
    save_mean = torch.empty(C, dtype=torch.float32, device=z.device)
    save_invstd = torch.empty(C, dtype=torch.float32, device=z.device)
    wsb = L.step_bn_train_workspace_bytes(M, C)
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=z.device)
    _capi.check(L.step_bn_train_forward(_dt(z), _lib.dptr(z), zcs, M, C, _lib.dptr(g_), _lib.dptr(b_), float(eps), float(momentum),
                                        _lib.dptr(running_mean), _lib.dptr(running_var), _lib.dptr(save_mean), _lib.dptr(save_invstd),
                                        int(bool(relu)), _lib.dptr(out), _chan_slice(out), _lib.dptr(ws), wsb, _lib.stream_ptr(z.device)),
                "step_bn_train_forward")
    return out, save_mean, save_invstd


def bn_train_backward(z, y, gy, gamma, save_mean, save_invstd, relu):
    """-> (gz dense in z.dtype, ggamma, gbeta fp32 [C])   step_bn_train_backward"""
    L = _lib.lib()
    C = z.shape[-1]
    M = z.numel() // C
    if gy.dtype not in (torch.float32, z.dtype):
        gy = gy.float()
    if gy.stride(-1) != 1:
        gy = gy.contiguous()
    try:
        gcs = _chan_slice(gy)
    except RuntimeError:
        gy = gy.contiguous()
        gcs = C
    gz = torch.empty(z.shape, dtype=z.dtype, device=z.device)
    ggamma = torch.empty(C, dtype=torch.float32, device=z.device)
    gbeta = torch.empty(C, dtype=torch.float32, device=z.device)
    g_ = None if gamma is None else (gamma if (gamma.dtype == torch.float32 and gamma.is_contiguous()) else gamma.float().contiguous())
    wsb = L.step_bn_train_workspace_bytes(M, C)
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=z.device)
    _capi.check(L.step_bn_train_backward(_dt(z), _lib.dptr(z), _chan_slice(z), _lib.dptr(y), _chan_slice(y), DT[gy.dtype], _lib.dptr(gy), gcs, M, C,
                                         int(bool(relu)), _lib.dptr(g_), _lib.dptr(save_mean), _lib.dptr(save_invstd), _lib.dptr(gz),
                                         _lib.dptr(ggamma), _lib.dptr(gbeta), _lib.dptr(ws), wsb, _lib.stream_ptr(z.device)),
                "step_bn_train_backward")
    return gz, ggamma, gbeta


def stem_forward(x, w_packed, Cout, scale, shift, out=None, relu=True):
    """x: [N,T,3,H,W] contiguous (the reference's input layout) -> channels-last [N,To,Ho,Wo,Cout]"""
    L = _lib.lib()
    N, T, C, H, W = x.shape
    if C != 3 or not x.is_contiguous():
        raise RuntimeError("step_amd: stem expects a contiguous [N,T,3,H,W] clip")
    To, Ho, Wo = (T - 2) // 2 + 1, (H - 2) // 2 + 1, (W - 2) // 2 + 1
    if out is None:
        out = torch.empty((N, To, Ho, Wo, Cout), dtype=x.dtype, device=x.device)
    def describe():
        pix = N * To * Ho * Wo
        buf = ctypes.create_string_buffer(256)
        _capi.check(L.step_stem_kernel_name(_dt(x), buf, 256), "step_stem_kernel_name")
        return buf.value.decode(), 2.0 * pix * Cout * 1029, (x.numel() + pix * Cout + Cout * 1029) * _ES[x.dtype]
    def launch():
        _capi.check(L.step_stem_forward(_dt(x), _lib.dptr(x), N, T, H, W, _lib.dptr(w_packed), _lib.dptr(scale), _lib.dptr(shift),
                                        int(bool(relu)), Cout, _lib.dptr(out), _chan_slice(out), 0, _lib.stream_ptr(x.device)), "step_stem_forward")
    _run(launch, describe)
    return out


def stem_pool_forward(x, w_packed, Cout, scale, shift):
    """The stem and the (1,3,3) / (1,2,2) max pool behind it as one call (step_stem_pool_forward): x [N,T,3,H,W] -> the POOLED
    channels-last tensor [N,To,Hp,Wp,Cout]; the un-pooled stem output never exists.  None when the library has no fused form for the
    shape / dtype (the caller runs the two layers one after the other; bit-identical)."""
    L = _lib.lib()
    N, T, C, H, W = x.shape
    if C != 3 or not x.is_contiguous() or x.dtype == torch.float32:
        return None
    wsb = L.step_stem_pool_workspace_bytes(_dt(x), N, T, H, W, Cout)
    if not wsb:
        return None
    To, Ho, Wo = (T - 2) // 2 + 1, (H - 2) // 2 + 1, (W - 2) // 2 + 1
    Hp, Wp = L.step_pool_out_size(Ho, 3, 2), L.step_pool_out_size(Wo, 3, 2)
    out = torch.empty((N, To, Hp, Wp, Cout), dtype=x.dtype, device=x.device)
    ws = torch.empty(wsb, dtype=torch.uint8, device=x.device)

    def describe():
        pix = N * To * Ho * Wo
        return ("void step::stem_stream_kernel<%s, 2, true, true, false>(step::StemParams)" % _TNAME[x.dtype], 2.0 * pix * Cout * 1029,
                (x.numel() + out.numel() + Cout * 1029) * _ES[x.dtype])

    def launch():
        # (the call in its two parts -- the stem's launch, then the seam pass -- so that the instrumented legs time them separately, as for the
        # fused conv3d_2c call: bench.py's roofline is the stem KERNEL's duration)
        _capi.check(L.step_stem_pool_forward_tiles(_dt(x), _lib.dptr(x), N, T, H, W, _lib.dptr(w_packed), _lib.dptr(scale), _lib.dptr(shift), Cout,
                                                   _lib.dptr(out), Cout, 0, _lib.dptr(ws), wsb, _lib.stream_ptr(x.device)), "step_stem_pool_forward_tiles")

    def finish():
        _capi.check(L.step_stem_pool_finish(_dt(x), _lib.dptr(x), N, T, H, W, Cout, _lib.dptr(out), Cout, 0, _lib.dptr(ws), wsb, _lib.stream_ptr(x.device)),
                    "step_stem_pool_finish")

    def describe_finish():
        th, tw = -(-Ho // 16), -(-Wo // 16)
        seam = N * To * ((th - 1) * Wp + (tw - 1) * Hp) * Cout
        rows = N * To * (th * Wo + tw * Ho) * Cout
        return ("step::stem_pool_fix_kernel(step::StemParams)", 0.0, (2 * seam + rows) * _ES[x.dtype])
    _run(launch, describe)
    _run(finish, describe_finish)
    return out


def stem_pool_forward_u8(frames, dtype, w_packed, Cout, scale, shift, u8_scale=2, mean=(0.0, 0.0, 0.0), std=(1.0, 1.0, 1.0)):
    """stem_pool_forward reading uint8 frames [N,T,H,W,3] directly (step_stem_pool_forward_u8): clip_from_u8's normalisation and the
    rounding to `dtype` happen in the stem's frame staging -- the normalised clip never exists.  Returns the pooled channels-last tensor
    [N,To,Hp,Wp,Cout] or None when the library has no fused form (the caller converts with clip_from_u8 first; bit-identical)."""
    L = _lib.lib()
    if frames.dtype != torch.uint8 or frames.dim() != 5 or frames.shape[-1] != 3 or not frames.is_contiguous() or not frames.is_cuda \
            or dtype not in (torch.bfloat16, torch.float16):
        return None
    N, T, H, W, _ = frames.shape
    code = _capi.BF16 if dtype == torch.bfloat16 else _capi.F16
    wsb = L.step_stem_pool_workspace_bytes(code, N, T, H, W, Cout)
    if not wsb:
        return None
    To, Ho, Wo = (T - 2) // 2 + 1, (H - 2) // 2 + 1, (W - 2) // 2 + 1
    Hp, Wp = L.step_pool_out_size(Ho, 3, 2), L.step_pool_out_size(Wo, 3, 2)
    out = torch.empty((N, To, Hp, Wp, Cout), dtype=dtype, device=frames.device)
    ws = torch.empty(wsb, dtype=torch.uint8, device=frames.device)
    m = (ctypes.c_float * 3)(*[float(v) for v in mean])
    sd = (ctypes.c_float * 3)(*[float(v) for v in std])

    def describe():
        pix = N * To * Ho * Wo
        return ("void step::stem_stream_kernel<%s, 2, true, true, true>(step::StemParams)" % _TNAME[dtype], 2.0 * pix * Cout * 1029,
                frames.numel() + (out.numel() + Cout * 1029) * _ES[dtype])

    def launch():
        _capi.check(L.step_stem_pool_forward_u8(code, _lib.dptr(frames), N, T, H, W, int(u8_scale), m, sd, _lib.dptr(w_packed), _lib.dptr(scale),
                                                _lib.dptr(shift), Cout, _lib.dptr(out), Cout, 0, _lib.dptr(ws), wsb, _lib.stream_ptr(frames.device)),
                    "step_stem_pool_forward_u8")
    _run(launch, describe)
    return out


def stem_wgrad(x, gy, Cout):
    """x [N,T,3,H,W] (the clip), gy fp32 channels-last [N,To,Ho,Wo,Cout] -> fp32 [Cout,3,7,7,7]"""
    L = _lib.lib()
    N, T, C, H, W = x.shape
    gy = gy.float().contiguous()
    dw = torch.empty((Cout, 3, 7, 7, 7), dtype=torch.float32, device=x.device)
    wsb = L.step_stem_wgrad_workspace_bytes(N, T, H, W, Cout) if WGRAD_WS else 0
    ws = torch.empty(wsb, dtype=torch.uint8, device=x.device) if wsb else None
    _capi.check(L.step_stem_wgrad_ws(_dt(x), _lib.dptr(x), N, T, H, W, _lib.dptr(gy), Cout, _lib.dptr(dw), 0, _lib.dptr(ws), wsb,
                                     _lib.stream_ptr(x.device)), "step_stem_wgrad_ws")
    return dw


def stem_wgrad16(x, gy, Cout):
    """stem_wgrad on the 16-bit matrix instructions: x [N,T,3,H,W] and gy channels-last [N,To,Ho,Wo,Cout] in the same 16-bit dtype ->
    fp32 [Cout,3,7,7,7]; None when the shape is outside the kernel's contract (the caller keeps stem_wgrad)."""
    L = _lib.lib()
    N, T, C, H, W = x.shape
    if gy.dtype != x.dtype or x.dtype == torch.float32:
        return None
    wsb = L.step_stem_wgrad16_workspace_bytes(_dt(x), N, T, H, W, Cout)
    if not wsb:
        return None
    x, gy = x.contiguous(), gy.contiguous()
    ws = torch.empty(wsb, dtype=torch.uint8, device=x.device)
    dw = torch.empty((Cout, 3, 7, 7, 7), dtype=torch.float32, device=x.device)
    prof = _NOPROF
    if PROFILE is not None:
        pix = gy.numel() // Cout
        prof = _Prof("void step::stem_wgrad16_kernel<%s>(step::StemWgrad16Params)" % _TNAME[x.dtype], 2.0 * pix * Cout * 1029,
                     (x.numel() + gy.numel()) * x.element_size())
    with prof:
        _capi.check(L.step_stem_wgrad16(_dt(x), _lib.dptr(x), N, T, H, W, _lib.dptr(gy), Cout, _lib.dptr(dw), 0, _lib.dptr(ws), wsb,
                                        _lib.stream_ptr(x.device)), "step_stem_wgrad16")
    return dw


def pool_out_size(L_, k, s):
    return _lib.lib().step_pool_out_size(L_, k, s)


def maxpool_tf(x, k, s, out=None):
    L = _lib.lib()
    N, D, H, W, C = x.shape
    if out is None:
        out = torch.empty((N, L.step_pool_out_size(D, k[0], s[0]), L.step_pool_out_size(H, k[1], s[1]),
                           L.step_pool_out_size(W, k[2], s[2]), C), dtype=x.dtype, device=x.device)
    def describe():
        ks = (tuple(k), tuple(s))
        sep = ks in (((3, 3, 3), (1, 1, 1)), ((1, 3, 3), (1, 2, 2)), ((3, 3, 3), (2, 2, 2))) and not _capi.get_option(L, "pool_direct")
        tn = _TNAME[x.dtype]
        kn = ("maxpool_sep_kernel<%s, %d, %d, %d, %d, %d, %d, 256>" % ((tn,) + ks[0] + ks[1]), ", int" * 7) if sep else \
            ("maxpool3d_tf_kernel<%s>" % tn, ", long long")
        return "void step::%s(%s const*, %s*, step::PoolParams%s)" % (kn[0], tn, tn, kn[1]), 0.0, (x.numel() + out.numel()) * _ES[x.dtype]
    def launch():
        _capi.check(L.step_maxpool3d_tf(_dt(x), _lib.dptr(x), N, D, H, W, C, _chan_slice(x), 0, k[0], k[1], k[2], s[0], s[1], s[2],
                                        _lib.dptr(out), _chan_slice(out), 0, _lib.stream_ptr(x.device)), "step_maxpool3d_tf")
    _run(launch, describe)
    return out


POOL_BWD_GATHER = True      # max-pool backward as two gathers (False: the fp32-atomic scatter form; module switch for tests / A-B timing)


def maxpool_tf_backward(x, gy, k, s):
    """x channels-last [N,D,H,W,C] (the pool's input), gy [N,Do,Ho,Wo,C] -> gx [N,D,H,W,C] (in x's dtype from the gather form, fp32
    from the atomic form that channel counts off the 16-byte grid fall back to)"""
    L = _lib.lib()
    N, D, H, W, C = x.shape
    vec = 16 // x.element_size()
    xcs = _chan_slice(x)
    if POOL_BWD_GATHER and C % vec == 0 and xcs % vec == 0 and x.data_ptr() % 16 == 0 and gy.dtype in (torch.float32, x.dtype):
        # two gathers, no atomics: gy as it arrives (fp32 or the activation type), gx written once in the activation type
        gy = gy.contiguous()
        gx = torch.empty((N, D, H, W, C), dtype=x.dtype, device=x.device)
        arg = torch.empty(gy.numel(), dtype=torch.uint8, device=x.device)
        _capi.check(L.step_maxpool3d_tf_backward_gather(_dt(x), _lib.dptr(x), N, D, H, W, C, xcs, 0, k[0], k[1], k[2], s[0], s[1], s[2],
                                                        _dt(gy), _lib.dptr(gy), _dt(x), _lib.dptr(gx), _lib.dptr(arg),
                                                        _lib.stream_ptr(x.device)), "step_maxpool3d_tf_backward_gather")
        return gx
    gy = gy.float().contiguous()
    gx = torch.empty((N, D, H, W, C), dtype=torch.float32, device=x.device)
    _capi.check(L.step_maxpool3d_tf_backward(_dt(x), _lib.dptr(x), N, D, H, W, C, xcs, 0, k[0], k[1], k[2], s[0], s[1], s[2],
                                             _lib.dptr(gy), _lib.dptr(gx), _lib.stream_ptr(x.device)), "step_maxpool3d_tf_backward")
    return gx


def clip_from_u8(frames, dtype=torch.float32, scale=2, mean=(0.0, 0.0, 0.0), std=(1.0, 1.0, 1.0), out=None):
    """uint8 frames [N,T,H,W,3] on the device -> normalised clip [N,T,3,H,W] (the reference's ConvertFromInts(scale) +
    SubtractMeans + DivideStds, data/augmentations.py:68-111, done after the PCIe transfer instead of before it).
    out: a contiguous [N,T,3,H,W] tensor to write (e.g. the static input of a captured step); its dtype wins."""
    L = _lib.lib()
    if frames.dtype != torch.uint8 or frames.dim() != 5 or frames.shape[-1] != 3 or not frames.is_contiguous():
        raise RuntimeError("step_amd: clip_from_u8 expects contiguous uint8 frames [N,T,H,W,3]")
    N, T, H, W, _ = frames.shape
    host = not frames.is_cuda
    if host:
        # ZERO-COPY ingest: page-locked host memory is mapped into the device's address space (hipHostMalloc), so the kernel can read
        # the frames over the host link itself -- the transfer and the conversion are one pass, no staging buffer, no copy engine,
        # no cross-stream event.  Only pinned tensors, and only with an explicit device-side `out`.
        if not frames.is_pinned() or out is None or not out.is_cuda:
            raise RuntimeError("step_amd: clip_from_u8 on host frames wants PINNED memory and a device tensor `out` (no CPU fallback)")
    if out is None:
        out = torch.empty((N, T, 3, H, W), dtype=dtype, device=frames.device)
    else:
        if tuple(out.shape) != (N, T, 3, H, W) or not out.is_contiguous() or (not host and out.device != frames.device):
            raise RuntimeError("step_amd: clip_from_u8(out=...) wants a contiguous [N,T,3,H,W] tensor on the frames' device")
        dtype = out.dtype
    m = (ctypes.c_float * 3)(*[float(v) for v in mean])
    sd = (ctypes.c_float * 3)(*[float(v) for v in std])
    code = {torch.float32: _capi.F32, torch.bfloat16: _capi.BF16, torch.float16: _capi.F16}[dtype]
    src = ctypes.c_void_p(frames.data_ptr()) if host else _lib.dptr(frames)
    _capi.check(L.step_clip_from_u8(src, N, T, H, W, int(scale), m, sd, code, _lib.dptr(out), _lib.stream_ptr(out.device)),
                "step_clip_from_u8")
    return out


def avgpool_hw(x, kh, kw):
    L = _lib.lib()
    N, D, H, W, C = x.shape
    if not x.is_contiguous():
        x = x.contiguous()
    out = torch.empty((N, D, H - kh + 1, W - kw + 1, C), dtype=x.dtype, device=x.device)
    _capi.check(L.step_avgpool_hw(_dt(x), _lib.dptr(x), N, D, H, W, C, kh, kw, _lib.dptr(out), _lib.stream_ptr(x.device)),
                "step_avgpool_hw")
    return out


def to_channels_last(x, dtype=None):
    """logical [N, C, *spatial] torch-contiguous -> [N, *spatial, C] contiguous (optionally cast)."""
    L = _lib.lib()
    x = x.contiguous()
    N, C = x.shape[0], x.shape[1]
    S = 1
    for v in x.shape[2:]:
        S *= v
    dtype = dtype or x.dtype
    out = torch.empty((N,) + tuple(x.shape[2:]) + (C,), dtype=dtype, device=x.device)
    _capi.check(L.step_transpose_cs(_lib.dptr(x), DT[x.dtype], _lib.dptr(out), DT[dtype], N, C, S, 1, _lib.stream_ptr(x.device)),
                "step_transpose_cs")
    return out


def from_channels_last(x, dtype=None):
    """[N, *spatial, C] contiguous -> torch-contiguous [N, C, *spatial]."""
    L = _lib.lib()
    x = x.contiguous()
    N, C = x.shape[0], x.shape[-1]
    S = 1
    for v in x.shape[1:-1]:
        S *= v
    dtype = dtype or x.dtype
    out = torch.empty((N, C) + tuple(x.shape[1:-1]), dtype=dtype, device=x.device)
    _capi.check(L.step_transpose_cs(_lib.dptr(x), DT[x.dtype], _lib.dptr(out), DT[dtype], N, C, S, 0, _lib.stream_ptr(x.device)),
                "step_transpose_cs")
    return out


# ------------------------------------------------------------------------------------------------
# ROI operators on raw tensors.  `layout`: NCHW for torch-contiguous maps, NHWC for channels-last.
def _roi_layout(inp):
    """inp: logical [B,C,H,W].  Returns (layout, dense tensor to hand to the kernel)."""
    if inp.dim() != 4:
        raise RuntimeError("step_amd: ROI ops expect a 4-D feature map [B,C,H,W]")
    if inp.is_contiguous():
        return _capi.NCHW, inp
    if inp.is_contiguous(memory_format=torch.channels_last) or inp.permute(0, 2, 3, 1).is_contiguous():
        return _capi.NHWC, inp
    return _capi.NCHW, inp.contiguous()


def _rois_f32(rois, device):
    r = rois.detach()
    if r.dtype != torch.float32:
        r = r.float()
    if r.device != device:
        r = r.to(device)
    return r.contiguous()


def roi_align_forward(inp, rois, ph, pw, scale, sampling_ratio):
    L = _lib.lib()
    layout, inp = _roi_layout(inp)
    B, C, H, W = inp.shape
    rois = _rois_f32(rois, inp.device)
    K = rois.shape[0]
    if layout == _capi.NHWC:
        out = torch.empty((K, ph, pw, C), dtype=inp.dtype, device=inp.device).permute(0, 3, 1, 2)
    else:
        out = torch.empty((K, C, ph, pw), dtype=inp.dtype, device=inp.device)
    _capi.check(L.step_roi_align_forward(_lib.dptr(inp), _dt(inp), layout, _lib.dptr(rois) if K else None, K, B, C, H, W,
                                         ph, pw, float(scale), int(sampling_ratio), _lib.dptr(out) if K else None,
                                         _lib.stream_ptr(inp.device)), "step_roi_align_forward")
    return out


def roi_align_tubes_forward(v, rois, ph, pw, scale, sampling_ratio):
    """ROIAlign over a T-slice of a channels-last feature buffer without copying it: v = buf[:, t0:t0+T] as a [B, T, H, W, C] view of a
    dense [B, T_all, H, W, C] buffer, rois [K,5] with first column b * T + t (frame of the slice).  step_roi_align_tubes_forward: no
    slice copy and no index arithmetic on the tubes.  Returns logical [K, C, ph, pw] (channels-last physical)."""
    L = _lib.lib()
    B, T, H, W, C = v.shape
    if not (v.stride(4) == 1 and v.stride(3) == C and v.stride(2) == W * C and v.stride(1) == H * W * C and v.stride(0) % v.stride(1) == 0):
        raise RuntimeError("step_amd: roi_align_tubes_forward wants a frame slice of a dense channels-last buffer")
    T_all = max(v.stride(0) // v.stride(1), T)
    rois = _rois_f32(rois, v.device)
    K = rois.shape[0]
    out = torch.empty((K, ph, pw, C), dtype=v.dtype, device=v.device)
    _capi.check(L.step_roi_align_tubes_forward(_lib.dptr(v), _dt(v), _lib.dptr(rois) if K else None, K, B, T_all, T, C, H, W, ph, pw,
                                               float(scale), int(sampling_ratio), _lib.dptr(out) if K else None,
                                               _lib.stream_ptr(v.device)), "step_roi_align_tubes_forward")
    return out.permute(0, 3, 1, 2)


def roi_align_backward(grad, rois, ph, pw, scale, sampling_ratio, B, C, H, W, deterministic=True):
    """deterministic=True: the fixed-order gather (bit-reproducible); False: the reference's fp32-atomics scatter
    (ROIAlign_cuda.cu:201-278).  A per-call argument of the C ABI (STEP_ROI_BWD_GATHER / _ATOMIC), not process state."""
    L = _lib.lib()
    g = grad.float()
    if g.is_contiguous():
        layout = _capi.NCHW
        gin = torch.empty((B, C, H, W), dtype=torch.float32, device=g.device)
    elif g.permute(0, 2, 3, 1).is_contiguous():
        layout = _capi.NHWC
        gin = torch.empty((B, H, W, C), dtype=torch.float32, device=g.device).permute(0, 3, 1, 2)
    else:
        layout, g = _capi.NCHW, g.contiguous()
        gin = torch.empty((B, C, H, W), dtype=torch.float32, device=g.device)
    rois = _rois_f32(rois, g.device)
    K = rois.shape[0]
    _capi.check(L.step_roi_align_backward(_lib.dptr(g) if K else None, layout, _lib.dptr(rois) if K else None, K, B, C, H, W,
                                          ph, pw, float(scale), int(sampling_ratio),
                                          _capi.ROI_BWD_GATHER if deterministic else _capi.ROI_BWD_ATOMIC, _lib.dptr(gin), _lib.stream_ptr(g.device)),
                "step_roi_align_backward")
    return gin.to(grad.dtype)


def roi_pool_forward(inp, rois, ph, pw, scale):
    L = _lib.lib()
    layout, inp = _roi_layout(inp)
    B, C, H, W = inp.shape
    rois = _rois_f32(rois, inp.device)
    K = rois.shape[0]
    if layout == _capi.NHWC:
        out = torch.empty((K, ph, pw, C), dtype=inp.dtype, device=inp.device).permute(0, 3, 1, 2)
        arg = torch.zeros((K, ph, pw, C), dtype=torch.int32, device=inp.device).permute(0, 3, 1, 2)
    else:
        out = torch.empty((K, C, ph, pw), dtype=inp.dtype, device=inp.device)
        arg = torch.zeros((K, C, ph, pw), dtype=torch.int32, device=inp.device)
    _capi.check(L.step_roi_pool_forward(_lib.dptr(inp), _dt(inp), layout, _lib.dptr(rois) if K else None, K, B, C, H, W, ph, pw,
                                        float(scale), _lib.dptr(out) if K else None, _lib.dptr(arg) if K else None,
                                        _lib.stream_ptr(inp.device)), "step_roi_pool_forward")
    return out, arg


def roi_pool_backward(grad, argmax, rois, ph, pw, B, C, H, W):
    L = _lib.lib()
    g = grad.float()
    nhwc = (not argmax.is_contiguous()) and argmax.permute(0, 2, 3, 1).is_contiguous()
    if nhwc:
        layout = _capi.NHWC
        if not g.permute(0, 2, 3, 1).is_contiguous():
            g = g.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
        gin = torch.empty((B, H, W, C), dtype=torch.float32, device=g.device).permute(0, 3, 1, 2)
    else:
        layout, g = _capi.NCHW, g.contiguous()
        gin = torch.empty((B, C, H, W), dtype=torch.float32, device=g.device)
    rois = _rois_f32(rois, g.device)
    K = rois.shape[0]
    _capi.check(L.step_roi_pool_backward(_lib.dptr(g) if K else None, _lib.dptr(argmax) if K else None, layout,
                                         _lib.dptr(rois) if K else None, K, B, C, H, W, ph, pw, _lib.dptr(gin),
                                         _lib.stream_ptr(g.device)), "step_roi_pool_backward")
    return gin.to(grad.dtype)


def nms_batched(boxes, scores, counts, threshold):
    """boxes [G,kmax,4], scores [G,kmax] (fp32, or fp64 for double inputs: the reference operator dispatches on the dtype,
    cpu/nms_cpu.cpp:95), counts [G] int32 (all on one device) -> keep mask uint8 [G,kmax] on the same device (no host round trip)."""
    L = _lib.lib()
    f64 = boxes.dtype == torch.float64
    dt = torch.float64 if f64 else torch.float32
    boxes, scores, counts = boxes.contiguous().to(dt), scores.contiguous().to(dt), counts.contiguous().to(torch.int32)
    G, kmax = scores.shape
    keep = torch.zeros((G, kmax), dtype=torch.uint8, device=boxes.device)
    if G == 0 or kmax == 0:
        return keep
    nb = L.step_nms_scratch_bytes(G, kmax)
    scratch = torch.empty(nb, dtype=torch.uint8, device=boxes.device) if nb else None
    fn = L.step_nms_batched_f64 if f64 else L.step_nms_batched
    _capi.check(fn(_lib.dptr(boxes), _lib.dptr(scores), _lib.dptr(counts), G, kmax, float(threshold),
                   _lib.dptr(keep), _lib.dptr(scratch), _lib.stream_ptr(boxes.device)), "step_nms_batched")
    return keep


def detect_nms(prob, loc, tube_start, tube_count, kmax, conf_thresh, nms_thresh, width, height, keep=None):
    """One refinement iteration's evaluation loop in one launch (step_detect_nms): prob [N,NC] and loc [N,4] fp32 (rows may be strided
    views), tube_start / tube_count [B] int32 -> (keep [B,NC,kmax] uint8 at the tubes' original slots, clamped boxes [N,4])."""
    L = _lib.lib()
    if prob.dtype != torch.float32 or loc.dtype != torch.float32 or prob.stride(1) != 1 or loc.stride(1) != 1:
        prob, loc = prob.float().contiguous(), loc.float().contiguous()
    N, NC = prob.shape
    B = tube_start.shape[0]
    if keep is None:
        keep = torch.empty((B, NC, kmax), dtype=torch.uint8, device=prob.device)
    boxes = torch.empty((N, 4), dtype=torch.float32, device=prob.device)
    _capi.check(L.step_detect_nms(_lib.dptr(prob), prob.stride(0), NC, _lib.dptr(loc), loc.stride(0), _lib.dptr(tube_start), _lib.dptr(tube_count),
                                  B, kmax, float(conf_thresh), float(nms_thresh), float(width), float(height), _lib.dptr(keep), _lib.dptr(boxes),
                                  _lib.stream_ptr(prob.device)), "step_detect_nms")
    return keep, boxes


def detect_compact(keep, boxes, scores, tube_start, width, height):
    """step_detect_compact: keep [I,B,NC,kmax] uint8, boxes / scores = lists of I tensors ([N,4] / [N,NC] fp32), tube_start [B] int32 ->
    (out_boxes [I*B*cap,4] normalised, out_scores, out_cls, out_tube, counts [I*B] int32), cap = NC * kmax: group g = i * B + b owns rows
    [g * cap, g * cap + counts[g]) in the reference's row order."""
    L = _lib.lib()
    I, B, NC, kmax = keep.shape
    dev = keep.device
    cap = NC * kmax
    boxes = [b if (b.dtype == torch.float32 and b.is_contiguous()) else b.float().contiguous() for b in boxes]
    scores = [s_ if (s_.dtype == torch.float32 and s_.stride(1) == 1) else s_.float().contiguous() for s_ in scores]
    ob = torch.empty((I * B * cap, 4), dtype=torch.float32, device=dev)
    os_ = torch.empty((I * B * cap,), dtype=torch.float32, device=dev)
    oc = torch.empty((I * B * cap,), dtype=torch.int64, device=dev)
    ot = torch.empty((I * B * cap,), dtype=torch.int64, device=dev)
    counts = torch.empty((I * B,), dtype=torch.int32, device=dev)
    bp = (ctypes.c_void_p * I)(*[b.data_ptr() for b in boxes])
    sp = (ctypes.c_void_p * I)(*[s_.data_ptr() for s_ in scores])
    ss = (ctypes.c_longlong * I)(*[s_.stride(0) for s_ in scores])
    _capi.check(L.step_detect_compact(_lib.dptr(keep), ctypes.cast(bp, ctypes.c_void_p), ctypes.cast(sp, ctypes.c_void_p), ctypes.cast(ss, ctypes.c_void_p),
                                      _lib.dptr(tube_start), I, B, NC, kmax, float(width), float(height), _lib.dptr(ob), _lib.dptr(os_), _lib.dptr(oc),
                                      _lib.dptr(ot), _lib.dptr(counts), _lib.stream_ptr(dev)), "step_detect_compact")
    return ob, os_, oc, ot, counts


def select_prepare(prob, loc, first, last, clip_of, gt_mid, gt_count, width, height):
    """step_select_prepare: prob [N,T,NC], loc [N,T,4], first / last [N,Tw,4] | None, clip_of [N] int32, gt_mid [B,Gmax,4], gt_count [B]
    int32 (one device) -> (mean_prob [N,NC], vloc [N,T,4], vfirst, vlast ([N,Tw,4] | None), iou [N,Gmax]), fp32 on that device."""
    L = _lib.lib()
    f32 = lambda t: None if t is None else t.detach().float().contiguous()
    prob, loc, first, last, gt_mid = f32(prob), f32(loc), f32(first), f32(last), f32(gt_mid)
    N, T, NC = prob.shape
    Tw = first.shape[1] if first is not None else 0
    Gmax = gt_mid.shape[1]
    dev = prob.device
    mean_prob = torch.empty((N, NC), dtype=torch.float32, device=dev)
    vloc = torch.empty((N, T, 4), dtype=torch.float32, device=dev)
    vfirst = torch.empty((N, Tw, 4), dtype=torch.float32, device=dev) if first is not None else None
    vlast = torch.empty((N, Tw, 4), dtype=torch.float32, device=dev) if first is not None else None
    iou = torch.empty((N, Gmax), dtype=torch.float32, device=dev)
    _capi.check(L.step_select_prepare(_lib.dptr(prob), _lib.dptr(loc), _lib.dptr(first), _lib.dptr(last), N, T, Tw, NC,
                                      _lib.dptr(clip_of.to(torch.int32).contiguous()), _lib.dptr(gt_mid), _lib.dptr(gt_count.to(torch.int32).contiguous()),
                                      Gmax, float(width), float(height), _lib.dptr(mean_prob), _lib.dptr(vloc), _lib.dptr(vfirst), _lib.dptr(vlast),
                                      _lib.dptr(iou), _lib.stream_ptr(dev)), "step_select_prepare")
    return mean_prob, vloc, vfirst, vlast, iou


def _check_head_targets(name, dev, N, Tl, NC, tubes, targets):
    """The kernel hard-codes tubes [N,Tl,5] and targets [N,3,6+NC] (row 2 = 'last'): anything else would be read out of bounds and give
    wrong losses silently, where the torch chain it replaces (two_branch.py:294-333) would have raised."""
    if targets is not None:
        if tuple(targets.shape) != (N, 3, 6 + NC):
            raise RuntimeError("step_amd: %s wants targets [N=%d, 3, 6+NC=%d], got %s" % (name, N, 6 + NC, tuple(targets.shape)))
        if tubes is None or tuple(tubes.shape) != (N, Tl, 5):
            raise RuntimeError("step_amd: %s wants tubes [N=%d, Tl=%d, 5], got %s" % (name, N, Tl, None if tubes is None else tuple(tubes.shape)))
        if targets.device != dev or tubes.device != dev:
            raise RuntimeError("step_amd: %s wants tubes and targets on %s" % (name, dev))
    elif tubes is not None and tuple(tubes.shape) != (N, Tl, 5):
        raise RuntimeError("step_amd: %s wants tubes [N=%d, Tl=%d, 5], got %s" % (name, N, Tl, tuple(tubes.shape)))


def head_outputs(logits, reg, N, Tl, T, NC, tubes=None, targets=None):
    """step_head_outputs: logits [N*Tl, ..., NC], reg [N*Tl, ..., 12] | None (2-D views of the GEMM outputs, activation dtype) ->
    (prob [N,NC], local_loc [N,Tl,4], first_loc, last_loc [N,T,4], loss_cls [N*NC | 1], loss_loc [1], loss_nbr [1]) fp32."""
    L = _lib.lib()
    dev = logits.device
    lg = logits.reshape(N * Tl, -1)
    rg = None if reg is None else reg.reshape(N * Tl, -1)
    if lg.stride(1) != 1 or (rg is not None and rg.stride(1) != 1):
        lg, rg = lg.contiguous(), (None if rg is None else rg.contiguous())
    f32 = lambda t: None if t is None else t.detach().float().contiguous()
    _check_head_targets("head_outputs", dev, N, Tl, NC, tubes, targets)
    if lg.shape[1] < NC or (rg is not None and rg.shape[1] < 12):
        raise RuntimeError("step_amd: head_outputs wants >= NC logit columns and >= 12 regression columns per row")
    tubes, targets = f32(tubes), f32(targets)
    train = targets is not None
    prob = torch.empty((N, NC), dtype=torch.float32, device=dev)
    ll = torch.empty((N, Tl, 4), dtype=torch.float32, device=dev) if rg is not None else None
    fl = torch.empty((N, T, 4), dtype=torch.float32, device=dev) if rg is not None else None
    la = torch.empty((N, T, 4), dtype=torch.float32, device=dev) if rg is not None else None
    losses = torch.empty((N * NC if train else 1) + 2, dtype=torch.float32, device=dev)
    ncls = N * NC if train else 1
    _capi.check(L.step_head_outputs(_dt(lg), _lib.dptr(lg), lg.stride(0), _lib.dptr(rg), 0 if rg is None else rg.stride(0), N, Tl, T, NC,
                                    _lib.dptr(tubes), _lib.dptr(targets), _lib.dptr(prob), _lib.dptr(ll), _lib.dptr(fl), _lib.dptr(la),
                                    _lib.dptr(losses), _lib.dptr(losses[ncls:]), _lib.dptr(losses[ncls + 1:]), _lib.stream_ptr(dev)),
                "step_head_outputs")
    return prob, ll, fl, la, losses[:ncls], losses[ncls:ncls + 1], losses[ncls + 1:ncls + 2], tubes, targets


def head_outputs_backward(logits, reg, N, Tl, T, NC, tubes, targets, g_cls, g_loc, g_nbr):
    """step_head_outputs_backward -> (g_logits, g_reg) shaped like logits / reg, in their dtype."""
    L = _lib.lib()
    lg = logits.reshape(N * Tl, -1)
    rg = None if reg is None else reg.reshape(N * Tl, -1)
    if lg.stride(1) != 1 or (rg is not None and rg.stride(1) != 1):
        lg, rg = lg.contiguous(), (None if rg is None else rg.contiguous())
    f32 = lambda t: None if t is None else t.detach().float().contiguous()
    _check_head_targets("head_outputs_backward", logits.device, N, Tl, NC, tubes, targets)
    if targets is None:
        raise RuntimeError("step_amd: head_outputs_backward wants the targets of the forward call")
    g_logits = torch.empty((N * Tl, NC), dtype=logits.dtype, device=logits.device)
    g_reg = torch.empty((N * Tl, 12), dtype=reg.dtype, device=reg.device) if rg is not None else None
    g_cls, g_loc, g_nbr = f32(g_cls), f32(g_loc), f32(g_nbr)
    _capi.check(L.step_head_outputs_backward(_dt(lg), _lib.dptr(lg), lg.stride(0), _lib.dptr(rg), 0 if rg is None else rg.stride(0), N, Tl, T, NC,
                                             _lib.dptr(tubes), _lib.dptr(targets), _lib.dptr(g_cls), _lib.dptr(g_loc), _lib.dptr(g_nbr),
                                             _lib.dptr(g_logits), _lib.dptr(g_reg), _lib.stream_ptr(logits.device)), "step_head_outputs_backward")
    return g_logits.reshape(logits.shape), (None if g_reg is None else g_reg.reshape(reg.shape))


def tube_update(flat, local_loc, first_loc, last_loc, clip_of, first_off, last_off, extend, width, height):
    """One refinement step's tube bookkeeping (step_tube_update): returns (pred_loc, pred_first, pred_last, next_flat)."""
    L = _lib.lib()
    N, T, _ = flat.shape
    Tw = first_loc.shape[1]
    f32 = lambda t: t.detach().float().contiguous()
    flat, local_loc, first_loc, last_loc = f32(flat), f32(local_loc), f32(first_loc), f32(last_loc)
    if clip_of.dtype != torch.int32:
        clip_of = clip_of.to(torch.int32)
    Tn = T + 2 * Tw if extend else T
    dev = flat.device
    pl = torch.empty((N, T, 4), dtype=torch.float32, device=dev)
    pf = torch.empty((N, Tw, 4), dtype=torch.float32, device=dev)
    pla = torch.empty((N, Tw, 4), dtype=torch.float32, device=dev)
    nxt = torch.empty((N, Tn, 5), dtype=torch.float32, device=dev)
    _capi.check(L.step_tube_update(_lib.dptr(flat), N, T, _lib.dptr(local_loc), _lib.dptr(first_loc), _lib.dptr(last_loc), Tw,
                                   int(first_off), int(last_off), _lib.dptr(clip_of.contiguous()), int(bool(extend)), float(width),
                                   float(height), _lib.dptr(pl), _lib.dptr(pf), _lib.dptr(pla), _lib.dptr(nxt),
                                   _lib.stream_ptr(dev)), "step_tube_update")
    return pl, pf, pla, nxt
