"""oracle -- CPU checkers for the STEP hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package, and only as the checker.  ``step_amd`` (the product) never imports it
and never falls back to it.

Contents
--------
``step_oracle.c`` / ``roi_nms``     plain-C restatement of ROIAlign / ROIPool / NMS
                                    (reference: external/maskrcnn_benchmark/csrc), loaded here
                                    through ctypes with numpy in / numpy out.
``i3d_ref.py``                      fp32 torch-CPU restatement of the I3D backbone, ContextNet and
                                    TwoBranchNet forward (reference: models/*.py).
``_ref/_C.so``                      the reference's own C++ CPU operators compiled from
                                    /root/reference where it lies (``make -C oracle ref``);
                                    ``load_reference_C()`` imports it.
"""
import ctypes
import importlib.util
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libstep_oracle.so")
_REF_PATH = os.path.join(_HERE, "_ref", "_C.so")
_lib = None


def build(ref=True):
    """Compile the C restatement (and the reference's own CPU ops when /root/reference exists)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "oracle"])
    if ref:
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build(ref=False)
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.orc_nms.restype = ctypes.c_int64
    return _lib


def have_reference_C():
    return os.path.exists(_REF_PATH)


def load_reference_C():
    """Import oracle/_ref/_C.so (the reference's pybind module, CPU build)."""
    import torch  # noqa: F401  (the extension links against libtorch)

    spec = importlib.util.spec_from_file_location("_C", _REF_PATH)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct))


def roi_align_forward(inp, rois, pooled, scale, sampling_ratio):
    """inp [B,C,H,W] f32, rois [K,5] -> [K,C,ph,pw]   (cpu/ROIAlign_cpu.cpp:137-243)"""
    inp, rois = _f32(inp), _f32(rois).reshape(-1, 5)
    B, C, H, W = inp.shape
    K = rois.shape[0]
    ph, pw = pooled
    out = np.empty((K, C, ph, pw), np.float32)
    if out.size:
        lib().orc_roi_align_forward(_p(inp, ctypes.c_float), _p(rois, ctypes.c_float), K, C, H, W, ph, pw,
                                    ctypes.c_float(scale), int(sampling_ratio), _p(out, ctypes.c_float))
    return out


def roi_align_backward(grad, rois, pooled, scale, sampling_ratio, in_shape):
    """grad [K,C,ph,pw] -> grad_input [B,C,H,W]   (cuda/ROIAlign_cuda.cu:201-278, serial)"""
    grad, rois = _f32(grad), _f32(rois).reshape(-1, 5)
    B, C, H, W = in_shape
    K = rois.shape[0]
    ph, pw = pooled
    gin = np.zeros((B, C, H, W), np.float32)
    if grad.size:
        lib().orc_roi_align_backward(_p(grad, ctypes.c_float), _p(rois, ctypes.c_float), K, B, C, H, W, ph, pw,
                                     ctypes.c_float(scale), int(sampling_ratio), _p(gin, ctypes.c_float))
    return gin


def roi_pool_forward(inp, rois, pooled, scale):
    """-> (out [K,C,ph,pw] f32, argmax int32)   (cuda/ROIPool_cuda.cu:40-101)"""
    inp, rois = _f32(inp), _f32(rois).reshape(-1, 5)
    B, C, H, W = inp.shape
    K = rois.shape[0]
    ph, pw = pooled
    out = np.empty((K, C, ph, pw), np.float32)
    arg = np.zeros((K, C, ph, pw), np.int32)
    if out.size:
        lib().orc_roi_pool_forward(_p(inp, ctypes.c_float), _p(rois, ctypes.c_float), K, C, H, W, ph, pw,
                                   ctypes.c_float(scale), _p(out, ctypes.c_float), _p(arg, ctypes.c_int32))
    return out, arg


def roi_pool_backward(grad, argmax, rois, pooled, in_shape):
    """(cuda/ROIPool_cuda.cu:103-132, serial)"""
    grad, rois = _f32(grad), _f32(rois).reshape(-1, 5)
    argmax = np.ascontiguousarray(argmax, dtype=np.int32)
    B, C, H, W = in_shape
    K = rois.shape[0]
    ph, pw = pooled
    gin = np.zeros((B, C, H, W), np.float32)
    if grad.size:
        lib().orc_roi_pool_backward(_p(grad, ctypes.c_float), _p(argmax, ctypes.c_int32), _p(rois, ctypes.c_float),
                                    K, B, C, H, W, ph, pw, _p(gin, ctypes.c_float))
    return gin


def nms(boxes, scores, thr):
    """boxes [n,4], scores [n] -> kept original indices ascending, int64   (cpu/nms_cpu.cpp:29-89)"""
    boxes, scores = _f32(boxes).reshape(-1, 4), _f32(scores).reshape(-1)
    n = boxes.shape[0]
    keep = np.empty((max(n, 1),), np.int64)
    m = lib().orc_nms(_p(boxes, ctypes.c_float), _p(scores, ctypes.c_float), ctypes.c_int64(n),
                      ctypes.c_float(thr), _p(keep, ctypes.c_int64))
    return keep[:m].copy()


def nms_f64(boxes, scores, thr):
    """the same for double-precision boxes / scores (the reference dispatches on the dtype, cpu/nms_cpu.cpp:95)"""
    boxes = np.ascontiguousarray(boxes, np.float64).reshape(-1, 4)
    scores = np.ascontiguousarray(scores, np.float64).reshape(-1)
    n = boxes.shape[0]
    keep = np.empty((max(n, 1),), np.int64)
    L = lib()
    L.orc_nms_f64.restype = ctypes.c_int64
    m = L.orc_nms_f64(_p(boxes, ctypes.c_double), _p(scores, ctypes.c_double), ctypes.c_int64(n), ctypes.c_float(thr), _p(keep, ctypes.c_int64))
    return keep[:m].copy()


def nms_batched(boxes, scores, counts, thr):
    """boxes [G,kmax,4], scores [G,kmax], counts [G] -> keep mask uint8 [G,kmax]"""
    boxes, scores = _f32(boxes), _f32(scores)
    counts = np.ascontiguousarray(counts, dtype=np.int32)
    G, kmax = scores.shape
    mask = np.zeros((G, kmax), np.uint8)
    if G and kmax:
        lib().orc_nms_batched(_p(boxes, ctypes.c_float), _p(scores, ctypes.c_float), _p(counts, ctypes.c_int32),
                              G, kmax, ctypes.c_float(thr), _p(mask, ctypes.c_uint8))
    return mask
