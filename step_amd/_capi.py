"""ctypes declarations of the C ABI in include/step_amd.h (one place, used by the product loader
step_amd/_lib.py and by the test-only emulation harness in tests/emul)."""
import ctypes as C

F32, BF16, F16 = 0, 1, 2
NCHW, NHWC = 0, 1
ROI_BWD_GATHER, ROI_BWD_ATOMIC = 0, 1
ABI_VERSION = 35

vp, fp, ip, u8p = C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p   # raw device addresses
i, f, ll, sz = C.c_int, C.c_float, C.c_longlong, C.c_size_t


class ConvDesc(C.Structure):
    """struct step_conv_desc (include/step_amd.h)"""
    _fields_ = [(n, C.c_int) for n in (
        "dtype", "N", "D", "H", "W", "Cin", "Cout", "kd", "kh", "kw", "x_cstride", "x_coff", "y_cstride", "y_coff",
        "res_cstride", "res_coff", "relu", "split", "y2_cstride", "y2_coff")]


class PackItem(C.Structure):
    """struct step_pack_item (include/step_amd.h)"""
    _fields_ = [("w", C.c_void_p), ("perm_c", C.c_void_p), ("packed", C.c_void_p)] + [(n, C.c_int) for n in (
        "Cout", "Cin", "w_cin", "cin_lo", "kd", "kh", "kw", "dgrad", "cin_pad", "reserved")]


class ConvItem(C.Structure):
    """struct step_conv_item (include/step_amd.h)"""
    _fields_ = [("desc", C.POINTER(ConvDesc)), ("x", C.c_void_p), ("w_packed", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p),
                ("res", C.c_void_p), ("y", C.c_void_p)]


class WgradReduceItem(C.Structure):
    """struct step_wgrad_reduce_item (include/step_amd.h)"""
    _fields_ = [("ws", C.c_void_p), ("dw", C.c_void_p), ("jobs", C.c_longlong)] + [(n, C.c_int) for n in (
        "kind", "gy", "nbw", "cot", "cit", "Cout", "Cin", "taps", "accumulate", "pw")]


WGRAD_REDUCE_MAX = 8

SIGNATURES = {
    "step_version": (C.c_char_p, []),
    "step_abi_version": (i, []),
    "step_set_option": (i, [i, i]),
    "step_get_option": (i, [i, C.POINTER(C.c_int)]),
    "step_reset_options": (None, []),
    "step_option_name": (C.c_char_p, [i]),
    "step_roi_align_forward": (i, [vp, i, i, fp, i, i, i, i, i, i, i, f, i, vp, vp]),
    "step_roi_align_tubes_forward": (i, [vp, i, fp, i, i, i, i, i, i, i, i, i, f, i, vp, vp]),
    "step_roi_align_backward": (i, [fp, i, fp, i, i, i, i, i, i, i, f, i, i, fp, vp]),
    "step_roi_pool_forward": (i, [vp, i, i, fp, i, i, i, i, i, i, i, f, vp, ip, vp]),
    "step_roi_pool_backward": (i, [fp, ip, i, fp, i, i, i, i, i, i, i, fp, vp]),
    "step_mfma_clock_probe": (i, [vp, i, i, vp]),
    "step_hbm_stream_probe": (i, [vp, vp, sz, i, vp]),
    "step_clock_sample": (i, [vp, i, i, vp]),
    "step_nms_scratch_bytes": (sz, [i, i]),
    "step_nms_batched": (i, [fp, fp, ip, i, i, f, u8p, vp, vp]),
    "step_nms_batched_f64": (i, [fp, fp, ip, i, i, f, u8p, vp, vp]),
    "step_detect_compact": (i, [u8p, vp, vp, vp, ip, i, i, i, i, f, f, fp, fp, vp, vp, ip, vp]),
    "step_detect_nms": (i, [fp, ll, i, fp, ll, ip, ip, i, i, f, f, f, f, u8p, fp, vp]),
    "step_conv_packed_elems": (sz, [i, i, i, i, i]),
    "step_conv_pack_weight": (i, [fp, i, i, i, i, i, i, ip, vp, vp]),
    "step_conv_pack_weight_dgrad": (i, [fp, i, i, i, i, i, i, i, vp, vp]),
    "step_conv_pack_weights": (i, [vp, i, i, vp]),
    "step_conv_forward": (i, [C.POINTER(ConvDesc), vp, vp, fp, fp, vp, vp, vp, vp]),
    "step_conv_forward_group": (i, [C.POINTER(ConvItem), i, vp]),
    "step_conv_group_kernel_name": (i, [C.POINTER(ConvItem), i, C.c_char_p, i]),
    "step_conv_workspace_bytes": (sz, [C.POINTER(ConvDesc)]),
    "step_pool_conv_forward": (i, [i, vp, i, i, i, i, i, i, i, vp, i, i, C.POINTER(ConvDesc), vp, vp, fp, fp, vp, vp, vp]),
    "step_pool_conv_plan_nb": (i, [C.POINTER(ConvDesc)]),
    "step_conv_forward_cat": (i, [C.POINTER(ConvDesc), vp, i, vp, i, i, vp, fp, fp, vp, vp, vp, vp]),
    "step_conv_forward_pre": (i, [C.POINTER(ConvDesc), vp, vp, fp, fp, vp, fp, fp, i, vp, vp]),
    "step_conv_pre_pool_workspace_bytes": (sz, [C.POINTER(ConvDesc)]),
    "step_conv_forward_pre_pool": (i, [C.POINTER(ConvDesc), vp, vp, fp, fp, vp, fp, fp, i, vp, vp, sz, vp]),
    "step_conv_forward_pre_pool_tiles": (i, [C.POINTER(ConvDesc), vp, vp, fp, fp, vp, fp, fp, i, vp, vp, sz, vp]),
    "step_conv_pre_pool_finish": (i, [C.POINTER(ConvDesc), vp, vp, sz, vp]),
    "step_conv_wgrad": (i, [C.POINTER(ConvDesc), vp, fp, fp, i, vp]),
    "step_conv_wgrad_workspace_bytes": (sz, [C.POINTER(ConvDesc)]),
    "step_conv_wgrad_ws": (i, [C.POINTER(ConvDesc), vp, fp, fp, i, vp, sz, vp]),
    "step_conv_wgrad16": (i, [C.POINTER(ConvDesc), vp, vp, fp, i, vp]),
    "step_conv_wgrad16_workspace_bytes": (sz, [C.POINTER(ConvDesc)]),
    "step_conv_wgrad16_ws": (i, [C.POINTER(ConvDesc), vp, vp, fp, i, vp, sz, vp]),
    "step_conv_wgrad_kernel_name": (i, [C.POINTER(ConvDesc), i, C.c_char_p, i]),
    "step_conv_wgrad_partial": (i, [C.POINTER(ConvDesc), vp, vp, i, fp, i, vp, sz, C.POINTER(WgradReduceItem), vp]),
    "step_wgrad_reduce_group": (i, [C.POINTER(WgradReduceItem), i, vp]),
    "step_conv_forward_ws": (i, [C.POINTER(ConvDesc), vp, vp, fp, fp, vp, vp, vp, vp, sz, vp]),
    "step_conv_kernel_name": (i, [C.POINTER(ConvDesc), C.c_char_p, i]),
    "step_conv_plan_info": (i, [C.POINTER(ConvDesc), C.POINTER(C.c_int), i]),
    "step_conv_pre_pool_plan_info": (i, [C.POINTER(ConvDesc), C.POINTER(C.c_int), i]),
    "step_stem_packed_elems": (sz, [i]),
    "step_stem_pack_weight": (i, [fp, i, i, vp, vp]),
    "step_stem_forward": (i, [i, vp, i, i, i, i, vp, fp, fp, i, i, vp, i, i, vp]),
    "step_stem_pool_workspace_bytes": (sz, [i, i, i, i, i, i]),
    "step_stem_pool_forward": (i, [i, vp, i, i, i, i, vp, fp, fp, i, vp, i, i, vp, sz, vp]),
    "step_stem_pool_forward_tiles": (i, [i, vp, i, i, i, i, vp, fp, fp, i, vp, i, i, vp, sz, vp]),
    "step_stem_pool_finish": (i, [i, vp, i, i, i, i, i, vp, i, i, vp, sz, vp]),
    "step_stem_pool_forward_u8": (i, [i, vp, i, i, i, i, i, C.POINTER(C.c_float), C.POINTER(C.c_float), vp, fp, fp, i, vp, i, i, vp, sz, vp]),
    "step_stem_kernel_name": (i, [i, C.c_char_p, i]),
    "step_stem_wgrad": (i, [i, vp, i, i, i, i, fp, i, fp, i, vp]),
    "step_stem_wgrad_workspace_bytes": (sz, [i, i, i, i, i]),
    "step_stem_wgrad_ws": (i, [i, vp, i, i, i, i, fp, i, fp, i, vp, sz, vp]),
    "step_stem_wgrad16_workspace_bytes": (sz, [i, i, i, i, i, i]),
    "step_stem_wgrad16": (i, [i, vp, i, i, i, i, vp, i, fp, i, vp, sz, vp]),
    "step_pool_out_size": (i, [i, i, i]),
    "step_maxpool3d_tf": (i, [i, vp, i, i, i, i, i, i, i, i, i, i, i, i, i, vp, i, i, vp]),
    "step_maxpool3d_tf_backward": (i, [i, vp, i, i, i, i, i, i, i, i, i, i, i, i, i, fp, fp, vp]),
    "step_maxpool3d_tf_backward_gather": (i, [i, vp, i, i, i, i, i, i, i, i, i, i, i, i, i, i, vp, i, vp, u8p, vp]),
    "step_clip_from_u8": (i, [vp, i, i, i, i, i, C.POINTER(C.c_float), C.POINTER(C.c_float), i, vp, vp]),
    "step_avgpool_hw": (i, [i, vp, i, i, i, i, i, i, i, vp, vp]),
    "step_transpose_cs": (i, [vp, i, vp, i, i, i, ll, i, vp]),
    "step_act_grad": (i, [i, vp, i, i, vp, i, fp, ll, i, i, fp, vp, vp]),
    "step_bn_train_workspace_bytes": (sz, [ll, i]),
    "step_bn_train_forward": (i, [i, vp, i, ll, i, fp, fp, f, f, fp, fp, fp, fp, i, vp, i, vp, sz, vp]),
    "step_bn_train_backward": (i, [i, vp, i, vp, i, i, vp, i, ll, i, i, fp, fp, fp, vp, fp, fp, vp, sz, vp]),
    "step_head_outputs": (i, [i, vp, i, vp, i, i, i, i, i, fp, fp, fp, fp, fp, fp, fp, fp, fp, vp]),
    "step_head_outputs_backward": (i, [i, vp, i, vp, i, i, i, i, i, fp, fp, fp, fp, fp, vp, vp, vp]),
    "step_tube_update": (i, [fp, i, i, fp, fp, fp, i, i, i, ip, i, f, f, fp, fp, fp, fp, vp]),
    "step_select_prepare": (i, [fp, fp, fp, fp, i, i, i, i, ip, fp, ip, i, f, f, fp, fp, fp, fp, fp, vp]),
    "step_adam_flat": (i, [fp, fp, fp, fp, ll, vp, fp, fp, i, C.c_double, C.c_double, C.c_double, i, f, i, vp]),
    "step_adam_flat_dev": (i, [fp, fp, fp, fp, ll, vp, fp, fp, i, C.c_double, C.c_double, C.c_double, vp, fp, f, i, vp]),
    "step_adam_flat_amp": (i, [fp, fp, fp, fp, ll, vp, fp, fp, i, C.c_double, C.c_double, C.c_double, vp, fp, f, i, fp, f, f, i, vp]),
}


def declare(lib, strict=True):
    """Attach argtypes/restype for every symbol of include/step_amd.h; raises AttributeError when the
    library does not export one of them (strict=False: tools that load an OLDER build next to the current one)."""
    for name, (res, args) in SIGNATURES.items():
        if not strict and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


_ERR = {-1: "unsupported dtype", -2: "bad shape", -3: "null pointer", -4: "unsupported configuration",
        -5: "misaligned pointer or channel count"}


def check(status, what):
    if status != 0:
        msg = _ERR.get(status, "hipError_t %d" % status) if status < 0 else "hipError_t %d" % status
        raise RuntimeError("%s failed: %s" % (what, msg))


# ---- planner options (include/step_amd.h: step_set_option) -----------------------------------------------------------
OPTION_IDS = {name: k for k, name in enumerate((
    "conv_impl", "conv_nb", "conv_waves", "conv_phased", "conv_gen", "conv_gmode", "conv_pws", "conv_splitk", "conv_tail",
    "conv_slots", "pool_direct", "wgrad_minpix", "wgrad16_lds", "conv_group_pw", "clip_vec", "conv_nb_rule", "throughput", "conv_persist", "conv_pws_waves"))}


def set_option(lib, name, value):
    check(lib.step_set_option(OPTION_IDS[name], int(value)), "step_set_option(%s, %r)" % (name, value))


def get_option(lib, name):
    v = C.c_int()
    check(lib.step_get_option(OPTION_IDS[name], C.byref(v)), "step_get_option(%s)" % name)
    return v.value


class options:
    """Context manager: `with options(lib, conv_waves=4, conv_nb=2): ...` sets planner options and restores the previous
    values on exit (tests and tools/ab_bench.py; the product never changes an option)."""

    def __init__(self, lib, **kw):
        self.lib, self.kw = lib, kw

    def __enter__(self):
        self.prev = {k: get_option(self.lib, k) for k in self.kw}
        for k, v in self.kw.items():
            set_option(self.lib, k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.prev.items():
            set_option(self.lib, k, v)
        return False
