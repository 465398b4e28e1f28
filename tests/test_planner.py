"""The launch planner of libstep_amd.so (host code, no GPU needed): step_conv_kernel_name / step_conv_workspace_bytes pin
which kernel family a layer shape is sent to, so a planner regression shows up on the CPU box."""
import ctypes
import os

import pytest

from step_amd import _capi

LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "step_amd", "libstep_amd.so")


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        pytest.skip("libstep_amd.so not built (python -c 'import __graft_entry__ as g; g.build()')")
    L = _capi.declare(ctypes.CDLL(LIB))
    L.step_reset_options()                  # the default dispatch
    return L


def name(L, dt, N, Cin, Cout, k, D, H, W):
    d = _capi.ConvDesc(dtype=dt, N=N, D=D, H=H, W=W, Cin=Cin, Cout=Cout, kd=k[0], kh=k[1], kw=k[2], x_cstride=Cin, x_coff=0, y_cstride=Cout,
                       y_coff=0, res_cstride=0, res_coff=0, relu=1, split=0, y2_cstride=0, y2_coff=0)
    buf = ctypes.create_string_buffer(256)
    assert L.step_conv_kernel_name(ctypes.byref(d), buf, 256) == 0
    return buf.value.decode(), L.step_conv_workspace_bytes(ctypes.byref(d))


def test_planner_dispatch(lib):
    BF, F32 = _capi.BF16, _capi.F32
    # conv3d_2c at C2: the pipelined kernel on 4-plane 8x8 tiles (56x56 maps tile exactly), 192 channels in one workgroup
    n, ws = name(lib, BF, 8, 64, 192, (3, 3, 3), 16, 56, 56)
    assert "conv_tap_kernel<step::bf16_t, 3, 3, 3, 3, 3, 2, 2, " in n and ws == 0
    # 400x400 clips: 50x50 maps get a general box (power-of-two tiles would waste 30 %)
    n, _ = name(lib, BF, 4, 128, 192, (3, 3, 3), 18, 50, 50)
    assert "conv_tap_kernel<step::bf16_t, 0," in n
    # the heads' 2-D convs: N folds into D, plane-folded general boxes on 7x7 maps
    n, _ = name(lib, BF, 132, 256, 256, (1, 3, 3), 1, 7, 7)
    assert "conv_tap_kernel<step::bf16_t, 0, " in n and ", 1, 3, 3," in n
    # deep pointwise layers stream through the 8-wave GEMM, shallow ones stay on the 4-wave kernel; K = 256 x many channel blocks
    # on a large map (the 3c fused triple) takes the weight-stationary stream, fp32 and residual layers never do
    assert "conv_pw_kernel" in name(lib, BF, 8, 480, 304, (1, 1, 1), 8, 14, 14)[0]
    # (3136 pixel groups on 2048 waves: the sixteen-wave form, at most one group per wave; the 56x56 map of the same layer keeps eight waves)
    assert "conv_pws_kernel<step::bf16_t, 2, 4, 16, false>" in name(lib, BF, 8, 256, 288, (1, 1, 1), 16, 28, 28)[0]
    assert "conv_pws_kernel<step::bf16_t, 3, 4, 8, false>" in name(lib, BF, 8, 256, 288, (1, 1, 1), 16, 56, 56)[0]
    assert "conv_pw_kernel" in name(lib, F32, 8, 256, 288, (1, 1, 1), 16, 28, 28)[0]
    assert "conv_igemm_kernel" in name(lib, BF, 8, 64, 64, (1, 1, 1), 16, 56, 56)[0]
    # few rows x very deep K (Linear 12544 -> 60 on 132 rows): split-K with a caller-owned workspace, fp32 included
    for dt in (BF, F32):
        n, ws = name(lib, dt, 132, 12544, 60, (1, 1, 1), 1, 1, 1)
        assert "pw_splitk_kernel" in n and ws > 0 and ws % 4 == 0
    # ... and on the 1224 rows of the last refinement step at the reference's 34 tubes per clip (4 clips x 34 tubes x 9 frames)
    assert "pw_splitk_kernel" in name(lib, BF, 1224, 12544, 60, (1, 1, 1), 1, 1, 1)[0]
    # tensors of >= 2^32 elements fall back to the 64-bit-offset kernel
    n, _ = name(lib, BF, 64, 64, 192, (3, 3, 3), 64, 512, 512)
    assert "conv_igemm_kernel" in n


def test_planner_nb_rule(lib):
    """conv_nb_rule (round 5 A/B, tools/plan_ab.py): 0 = the accumulator depth that needs the fewest ROUNDS of the chip (the default: the
    14x14 layers take NB = 1, one round of 224-280 workgroups), 1 = the depth with the least chip time (workgroups x per-workgroup time):
    the 14x14 layers go deeper, conv3d_2c and the 28x28 layers keep their depth."""
    BF = _capi.BF16
    nb = lambda n: int(n.split("conv_tap_kernel<step::bf16_t, ")[1].split(",")[1])
    n4d0, _ = name(lib, BF, 8, 128, 256, (3, 3, 3), 8, 14, 14)
    n2c0, _ = name(lib, BF, 8, 64, 192, (3, 3, 3), 16, 56, 56)
    n3c0, _ = name(lib, BF, 8, 128, 192, (3, 3, 3), 16, 28, 28)
    assert nb(n4d0) == 1 and nb(n2c0) == 3 and nb(n3c0) == 3
    with _capi.options(lib, conv_nb_rule=1):
        assert nb(name(lib, BF, 8, 128, 256, (3, 3, 3), 8, 14, 14)[0]) == 2
        assert nb(name(lib, BF, 8, 160, 320, (3, 3, 3), 8, 14, 14)[0]) == 3
        assert name(lib, BF, 8, 64, 192, (3, 3, 3), 16, 56, 56)[0] == n2c0
        assert name(lib, BF, 8, 128, 192, (3, 3, 3), 16, 28, 28)[0] == n3c0
    assert name(lib, BF, 8, 128, 256, (3, 3, 3), 8, 14, 14)[0] == n4d0


def test_planner_pooled_conv_replans_onto_tiles(lib):
    """step_conv_forward_pre_pool exists where the map is (or may be) tiled 4 x 8 x 8: C2's 56 x 56 maps by the planner's own choice; the
    100 x 100 x 18-plane maps of AVA clips by re-planning (the planner alone prefers 720 general boxes to 845 tiles: within the pooled call's
    25 % allowance, and faster with the pool in the epilogue, tools/prepool_ab.py); not the 14 x 14 maps, not without ReLU, not fp32."""
    BF = _capi.BF16

    def desc(N, D, H, W, Cout=192, relu=1, dt=BF):
        return _capi.ConvDesc(dtype=dt, N=N, D=D, H=H, W=W, Cin=64, Cout=Cout, kd=3, kh=3, kw=3, x_cstride=64, x_coff=0, y_cstride=Cout, y_coff=0,
                              res_cstride=0, res_coff=0, relu=relu, split=0, y2_cstride=0, y2_coff=0)
    info = (ctypes.c_int * 10)()
    d = desc(8, 16, 56, 56)
    assert lib.step_conv_plan_info(ctypes.byref(d), info, 10) == 0 and info[1] == 3
    assert lib.step_conv_pre_pool_workspace_bytes(ctypes.byref(d)) == 8 * 16 * (7 * 56 + 7 * 56) * 192 * 2
    d = desc(4, 18, 100, 100)
    assert lib.step_conv_plan_info(ctypes.byref(d), info, 10) == 0 and info[1] == 0 and info[9] == 4 * 720
    assert lib.step_conv_pre_pool_workspace_bytes(ctypes.byref(d)) == 4 * 18 * (13 * 100 + 13 * 100) * 192 * 2
    assert lib.step_conv_pre_pool_workspace_bytes(ctypes.byref(desc(8, 8, 14, 14))) == 0
    assert lib.step_conv_pre_pool_workspace_bytes(ctypes.byref(desc(8, 16, 56, 56, relu=0))) == 0
    assert lib.step_conv_pre_pool_workspace_bytes(ctypes.byref(desc(8, 16, 56, 56, dt=_capi.F32))) == 0


def wgrad_name(L, dt, N, Cin, Cout, k, D, H, W, xcs=None):
    d = _capi.ConvDesc(dtype=dt, N=N, D=D, H=H, W=W, Cin=Cin, Cout=Cout, kd=k[0], kh=k[1], kw=k[2], x_cstride=xcs or Cin, x_coff=0, y_cstride=Cout,
                       y_coff=0, res_cstride=0, res_coff=0, relu=0, split=0, y2_cstride=0, y2_coff=0)
    buf = ctypes.create_string_buffer(256)
    assert L.step_conv_wgrad_kernel_name(ctypes.byref(d), 1, buf, 256) == 0
    return buf.value.decode(), L.step_conv_wgrad16_workspace_bytes(ctypes.byref(d))


def test_weight_gradient_planner(lib):
    """Which 16-bit weight-gradient form a layer of the training step gets (round 4), and the workspace each form asks for."""
    BF = _capi.BF16
    # 3x3x3 windows: the twelve-wave LDS-tiled kernel; 100-wide maps (one clip or eight) the wide-halo instantiation
    n, ws = wgrad_name(lib, BF, 8, 128, 192, (3, 3, 3), 18, 50, 50)
    assert "conv_wgrad16_lds12_kernel<step::bf16_t, 3>" in n and ws > 0 and ws % (144 * 64 * 16) == 0      # whole workgroup blocks of partial tiles
    n, _ = wgrad_name(lib, BF, 1, 64, 192, (3, 3, 3), 18, 100, 100)
    assert "conv_wgrad16_lds12_wide_kernel" in n
    n, _ = wgrad_name(lib, BF, 8, 64, 192, (3, 3, 3), 16, 56, 56)
    assert "conv_wgrad16_lds12_wide_kernel" in n                                # four rows per chunk instead of three
    n, _ = wgrad_name(lib, BF, 120, 256, 256, (1, 3, 3), 9, 7, 7)
    assert "conv_wgrad16_lds12_kernel" in n
    # pointwise layers with >= 2048 pixels: the pixel stream, dense [Cout, Cin] fp32 images per slice
    for (N, Cin, Cout, D, H, W) in ((8, 192, 96, 18, 50, 50), (1, 480, 304, 9, 25, 25), (1080, 832, 624, 1, 7, 7), (1, 64, 64, 18, 100, 100)):
        n, ws = wgrad_name(lib, BF, N, Cin, Cout, (1, 1, 1), D, H, W)
        assert "conv_wgrad16_pws_kernel<step::bf16_t>" in n, (N, Cin, Cout, n)
        assert ws > 0 and ws % (Cout * Cin * 4) == 0 and ws // (Cout * Cin * 4) <= 4096, (N, Cin, Cout, ws)
    # a channel-sliced input keeps the form; fewer pixels fall back to the tiled / per-tap kernels; odd channel counts to the per-tap one
    assert "conv_wgrad16_pws_kernel" in wgrad_name(lib, BF, 8, 192, 96, (1, 1, 1), 18, 50, 50, xcs=256)[0]
    assert "conv_wgrad16_lds2_kernel<step::bf16_t, 1>" in wgrad_name(lib, BF, 1, 480, 304, (1, 1, 1), 4, 13, 13)[0]
    assert "conv_wgrad_kernel<step::bf16_t, 2, 2, true>" in wgrad_name(lib, BF, 1, 192, 96, (1, 1, 1), 3, 13, 13)[0]
    assert "conv_wgrad_kernel" in wgrad_name(lib, BF, 8, 192, 81, (1, 1, 1), 18, 50, 50)[0]
    # option 2 keeps the pointwise layers on the earlier forms (A/B, tests)
    _capi.set_option(lib, "wgrad16_lds", 2)
    try:
        assert "conv_wgrad_kernel" in wgrad_name(lib, BF, 8, 192, 96, (1, 1, 1), 18, 50, 50)[0]
        assert "conv_wgrad16_lds2_kernel" in wgrad_name(lib, BF, 1080, 832, 624, (1, 1, 1), 1, 7, 7)[0]
        assert "conv_wgrad16_lds12_kernel" in wgrad_name(lib, BF, 8, 128, 192, (3, 3, 3), 18, 50, 50)[0]
    finally:
        _capi.set_option(lib, "wgrad16_lds", 1)


def test_abi_and_symbols(lib):
    """Every entry point include/step_amd.h declares is exported by the library and bound in step_amd/_capi.py (and nothing
    is bound that the header does not declare); no compute call is made."""
    import os
    import re

    assert lib.step_abi_version() == _capi.ABI_VERSION
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "step_amd.h")).read()
    declared = set(re.findall(r"STEP_API\s+[\w\s\*]+?\b(step_\w+)\s*\(", hdr))
    assert len(declared) >= 29, sorted(declared)
    for sym in sorted(declared):
        assert hasattr(lib, sym), "declared in the header but not exported: " + sym
    assert declared == set(_capi.SIGNATURES), (sorted(declared - set(_capi.SIGNATURES)), sorted(set(_capi.SIGNATURES) - declared))


def test_options_api_and_no_environment(lib):
    """step_set_option / step_get_option: names and ids agree with step_amd/_capi.py, range checks, the context manager restores,
    an option really moves the planner -- and the library source reads no environment variable."""
    import glob

    for nm, k in _capi.OPTION_IDS.items():
        assert lib.step_option_name(k).decode() == nm
    assert lib.step_option_name(len(_capi.OPTION_IDS)) is None
    assert lib.step_set_option(999, 1) == -2 and lib.step_set_option(_capi.OPTION_IDS["conv_waves"], 5) == -2
    BF = _capi.BF16
    base, _ = name(lib, BF, 8, 96, 208, (3, 3, 3), 8, 14, 14)
    assert base.endswith(", 2, 2, 8, 1>(step::ConvParams)"), base
    with _capi.options(lib, conv_phased=0, conv_waves=4):
        assert _capi.get_option(lib, "conv_phased") == 0
        n4, _ = name(lib, BF, 8, 96, 208, (3, 3, 3), 8, 14, 14)
        assert n4.endswith(", 2, 4, 0>(step::ConvParams)"), n4
    assert _capi.get_option(lib, "conv_phased") == 2 and _capi.get_option(lib, "conv_waves") == 0
    assert name(lib, BF, 8, 96, 208, (3, 3, 3), 8, 14, 14)[0] == base
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for f in glob.glob(os.path.join(root, "step_amd", "csrc", "*.h*")):
        assert "getenv" not in open(f).read(), f
