"""INTEGRATION.md Option B on the GPU: integration/_C.py (the ctypes binding that replaces the reference's pybind extension) over the
real libstep_amd.so -- the five `_C` entry points with the pybind signatures against the C restatement."""
import importlib.util
import os

import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _shim():
    spec = importlib.util.spec_from_file_location("step_option_b_C", os.path.join(ROOT, "integration", "_C.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_option_b_shim_matches_the_oracle():
    C_ = _shim()
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(9)
    B, C, H, W = 2, 40, 25, 25
    x = rs.randn(B, C, H, W).astype(np.float32)
    rois = np.array([[0, 0, 0, 399, 399], [1, 33.3, 50.1, 180.7, 222.2], [1, 300, 120, 399, 380], [0, 10, 10, 12, 11]], np.float32)
    g = rs.randn(rois.shape[0], C, 7, 7).astype(np.float32)
    xt, rt, gt = torch.from_numpy(x).to(dev), torch.from_numpy(rois).to(dev), torch.from_numpy(g).to(dev)
    y = C_.roi_align_forward(xt, rt, 1 / 16., 7, 7, 0)
    assert np.array_equal(y.cpu().numpy(), oracle.roi_align_forward(x, rois, (7, 7), 1 / 16., 0))
    gi = C_.roi_align_backward(gt, rt, 1 / 16., 7, 7, B, C, H, W, 0)
    ref = oracle.roi_align_backward(g, rois, (7, 7), 1 / 16., 0, x.shape)
    assert np.abs(gi.cpu().numpy() - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())
    out, arg = C_.roi_pool_forward(xt, rt, 1 / 16., 7, 7)
    ro, ra = oracle.roi_pool_forward(x, rois, (7, 7), 1 / 16.)
    assert np.array_equal(out.cpu().numpy(), ro) and np.array_equal(arg.cpu().numpy(), ra) and arg.dtype == torch.int32
    gp = C_.roi_pool_backward(gt, xt, rt, arg, 1 / 16., 7, 7, B, C, H, W)
    assert np.abs(gp.cpu().numpy() - oracle.roi_pool_backward(g, ra, rois, (7, 7), x.shape)).max() <= 1e-5
    for n in (1, 34, 200):
        xy = rs.uniform(0, 300, (n, 2))
        boxes = np.concatenate([xy, xy + rs.uniform(10, 150, (n, 2))], 1)
        scores = rs.permutation(n).astype(np.float64) / n
        k = C_.nms(torch.from_numpy(boxes.astype(np.float32)), torch.from_numpy(scores.astype(np.float32)), 0.4)     # CPU inputs, as test.py:158-161 passes them
        assert k.dtype == torch.int64 and k.device.type == "cpu"
        assert np.array_equal(k.numpy(), oracle.nms(boxes.astype(np.float32), scores.astype(np.float32), 0.4))
        k64 = C_.nms(torch.from_numpy(boxes).to(dev), torch.from_numpy(scores).to(dev), 0.4)
        assert np.array_equal(k64.numpy(), oracle.nms_f64(boxes, scores, 0.4))
    assert C_.nms(torch.zeros(0, 4), torch.zeros(0), 0.4).numel() == 0
