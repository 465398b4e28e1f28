"""tools/roi_bwd_bench.py -- ROIAlign backward (fixed-order gather | atomics) at the training step's shapes (GPU only, tuning aid)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes
from step_amd import _capi, _lib, ops  # noqa: E402
from step_amd.tube_math import generate_anchors  # noqa: E402

for clips, tubes, Tl in ((1, 5, 9), (8, 15, 3), (8, 15, 9), (8, 34, 9)):
    a = torch.from_numpy(generate_anchors()[:tubes] * 400.0).float()
    rois = []
    for b in range(clips):
        for k in range(tubes):
            for t in range(Tl):
                rois.append([b * Tl + t] + a[k].tolist())
    rois = torch.tensor(rois, device="cuda")
    K = rois.shape[0]
    g = torch.randn(K, 7, 7, 832, device="cuda").permute(0, 3, 1, 2)
    out = {}
    libs = {"cur": _lib.lib()}
    pp = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libstep_amd_prev.so")
    if os.path.exists(pp):
        libs["prev"] = _capi.declare(ctypes.CDLL(pp), strict=False)
    for nm, L_ in libs.items():
        _lib._LIB = L_
        ops.roi_align_backward(g, rois, 7, 7, 1 / 16., 0, clips * Tl, 832, 25, 25)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            ops.roi_align_backward(g, rois, 7, 7, 1 / 16., 0, clips * Tl, 832, 25, 25)
        torch.cuda.synchronize()
        print("   lib=%s gather %.3f ms" % (nm, (time.perf_counter() - t0) / 5 * 1e3))
    _lib._LIB = libs["cur"]
    for det in (True, False):
        r = ops.roi_align_backward(g, rois, 7, 7, 1 / 16., 0, clips * Tl, 832, 25, 25, deterministic=det)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            ops.roi_align_backward(g, rois, 7, 7, 1 / 16., 0, clips * Tl, 832, 25, 25, deterministic=det)
        torch.cuda.synchronize()
        out[det] = ((time.perf_counter() - t0) / 5 * 1e3, r)
    err = float((out[True][1] - out[False][1]).abs().max() / out[False][1].abs().max())
    print("clips %d tubes %2d Tl %d  K %5d:  gather %.3f ms   atomics %.3f ms   (max rel diff %.1e)" % (clips, tubes, Tl, K, out[True][0], out[False][0], err))
