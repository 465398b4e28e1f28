"""integration/_C.py -- INTEGRATION.md Option B as a file: the binding a maintainer of the reference drops in as
`external/maskrcnn_benchmark/roi_layers/_C.py` INSTEAD of building the pybind11 extension of setup.py (csrc/vision.cpp:30-36).

The reference's own Python (roi_layers/{nms,roi_align,roi_pool}.py: autograd Functions, modules, apex float_function) stays as
it is and calls these five functions with the pybind signatures of csrc/nms.h:34-36, ROIAlign.h:35-60, ROIPool.h:35-60; they call
the C ABI of libstep_amd.so (include/step_amd.h) through ctypes.  Tested artefact: oracle/check_option_b.py runs the reference's
roi_layers over this file in the build container, tests/test_gpu_option_b.py runs it on the GPU.

The library is looked up next to the step_amd package unless STEP_AMD_LIB names another file.  There is no CPU path: tensors that
are not on a ROCm device make the launch fail (the build-container check points STEP_AMD_LIB at the host interpreter build).
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_L = ctypes.CDLL(os.environ.get("STEP_AMD_LIB") or os.path.join(os.path.dirname(_HERE), "step_amd", "libstep_amd.so"))
_vp, _i, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
_L.step_roi_align_forward.argtypes = [_vp, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp, _vp]
_L.step_roi_align_backward.argtypes = [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _i, _i, _vp, _vp]
_L.step_roi_pool_forward.argtypes = [_vp, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp]
_L.step_roi_pool_backward.argtypes = [_vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]
_L.step_nms_scratch_bytes.restype = ctypes.c_size_t
_L.step_nms_scratch_bytes.argtypes = [_i, _i]
_L.step_nms_batched.argtypes = [_vp, _vp, _vp, _i, _i, _f, _vp, _vp, _vp]
_L.step_nms_batched_f64.argtypes = [_vp, _vp, _vp, _i, _i, _f, _vp, _vp, _vp]
_DT = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}
_NCHW = 0
# ROIAlign backward mode (a per-call argument of the C ABI): 0 = fixed-order gather (bit-reproducible; every cell walks the K roi
# headers), 1 = the algorithm of cuda/ROIAlign_cuda.cu:201-278 (zero + fp32 atomics).  The shim stands in for the reference's own
# CUDA operator on torch-contiguous maps with hundreds of rois, so it runs that algorithm unless STEP_ROI_BWD_DETERMINISTIC=1.
_ROI_BWD_MODE = 0 if os.environ.get("STEP_ROI_BWD_DETERMINISTIC") == "1" else 1


def _p(t):
    return None if t is None or t.numel() == 0 else ctypes.c_void_p(t.data_ptr())


def _s(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream) if t.is_cuda else None


def _chk(rc, what):
    if rc:
        raise RuntimeError("%s failed (%d)" % (what, rc))             # the pybind ops throw through AT_ASSERTM / THCudaCheck


def roi_align_forward(input, rois, spatial_scale, pooled_height, pooled_width, sampling_ratio):        # csrc/ROIAlign.h:35-40
    input, rois = input.contiguous(), rois.contiguous().float()
    B, C, H, W = input.shape
    K = rois.shape[0]
    out = torch.empty((K, C, pooled_height, pooled_width), dtype=input.dtype, device=input.device)
    _chk(_L.step_roi_align_forward(_p(input), _DT[input.dtype], _NCHW, _p(rois), K, B, C, H, W, pooled_height, pooled_width,
                                   spatial_scale, sampling_ratio, _p(out), _s(input)), "roi_align_forward")
    return out


def roi_align_backward(grad, rois, spatial_scale, pooled_height, pooled_width, batch_size, channels, height, width, sampling_ratio):   # ROIAlign.h:51-60
    grad, rois = grad.contiguous().float(), rois.contiguous().float()
    gin = torch.empty((batch_size, channels, height, width), dtype=torch.float32, device=grad.device)     # zeroed by the op
    _chk(_L.step_roi_align_backward(_p(grad), _NCHW, _p(rois), rois.shape[0], batch_size, channels, height, width, pooled_height,
                                    pooled_width, spatial_scale, sampling_ratio, _ROI_BWD_MODE, _p(gin), _s(grad)), "roi_align_backward")
    return gin


def roi_pool_forward(input, rois, spatial_scale, pooled_height, pooled_width):                           # csrc/ROIPool.h:35-39
    input, rois = input.contiguous(), rois.contiguous().float()
    B, C, H, W = input.shape
    K = rois.shape[0]
    out = torch.empty((K, C, pooled_height, pooled_width), dtype=input.dtype, device=input.device)
    arg = torch.zeros((K, C, pooled_height, pooled_width), dtype=torch.int32, device=input.device)
    _chk(_L.step_roi_pool_forward(_p(input), _DT[input.dtype], _NCHW, _p(rois), K, B, C, H, W, pooled_height, pooled_width,
                                  spatial_scale, _p(out), _p(arg), _s(input)), "roi_pool_forward")
    return out, arg


def roi_pool_backward(grad, input, rois, argmax, spatial_scale, pooled_height, pooled_width, batch_size, channels, height, width):   # ROIPool.h:50-60
    grad, rois = grad.contiguous().float(), rois.contiguous().float()
    gin = torch.empty((batch_size, channels, height, width), dtype=torch.float32, device=grad.device)
    _chk(_L.step_roi_pool_backward(_p(grad), _p(argmax.contiguous()), _NCHW, _p(rois), rois.shape[0], batch_size, channels, height, width,
                                   pooled_height, pooled_width, _p(gin), _s(grad)), "roi_pool_backward")
    return gin


_NMS_COUNTS = {}


def nms(dets, scores, threshold):                                                                        # csrc/nms.h:34-36
    if dets.numel() == 0:
        return torch.empty((0,), dtype=torch.int64)                                                      # nms.h:41-42
    f64 = dets.dtype == torch.float64                                                                    # AT_DISPATCH_FLOATING_TYPES, nms_cpu.cpp:95
    dt = torch.float64 if f64 else torch.float32
    n = dets.shape[0]
    if dets.is_cuda or not torch.cuda.is_available():
        dev = dets.device
        d, s = dets.to(dtype=dt).contiguous(), scores.to(device=dev, dtype=dt).contiguous()
    else:                                                                                                # CPU tensors (test.py:158-160): one upload [4n | n]
        dev = torch.device("cuda", torch.cuda.current_device())
        buf = torch.cat([dets.to(dt).reshape(-1), scores.to(device="cpu", dtype=dt).reshape(-1)]).to(dev)
        d, s = buf[:4 * n], buf[4 * n:]
    cnt = _NMS_COUNTS.get((dev, n))
    if cnt is None:
        cnt = _NMS_COUNTS[(dev, n)] = torch.tensor([n], dtype=torch.int32, device=dev)
    keep = torch.zeros((1, n), dtype=torch.uint8, device=dev)
    scratch = torch.empty(max(_L.step_nms_scratch_bytes(1, n), 16), dtype=torch.uint8, device=dev)
    fn = _L.step_nms_batched_f64 if f64 else _L.step_nms_batched
    _chk(fn(_p(d), _p(s), _p(cnt), 1, n, threshold, _p(keep), _p(scratch), _s(d)), "nms")
    return torch.nonzero(keep[0].cpu()).squeeze(1)                                                       # ascending original indices (nms_cpu.cpp:88); the mask comes down in one copy
