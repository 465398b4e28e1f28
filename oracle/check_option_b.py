"""oracle/check_option_b.py -- BUILD-CONTAINER ONLY: INTEGRATION.md Option B as a running thing.

The reference's OWN operator layer -- external/maskrcnn_benchmark/roi_layers/{__init__,nms,roi_align,roi_pool}.py: the autograd
Functions _ROIAlign / _ROIPool, the modules, nms -- is imported UNCHANGED from /root/reference with integration/_C.py standing in for
its pybind extension (registered as `external.maskrcnn_benchmark.roi_layers._C`, which is where a maintainer drops the file), the
C ABI served by the host interpreter build of the kernels.  Forward, backward through torch autograd and nms are compared with the
C restatement (oracle/step_oracle.c: bit-identical to the reference's own CPU operators where those exist).

    python -m oracle.check_option_b
"""
import importlib.util
import os
import subprocess
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("STEP_REFERENCE", "/root/reference")


def main():
    if not os.path.isdir(REF):
        raise SystemExit("the reference tree is not available (%s): build-container check only" % REF)
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emul"), "-j8"])
    os.environ["STEP_AMD_LIB"] = os.path.join(ROOT, "tests", "emul", "_build", "libstep_amd_emul.so")
    sys.path.insert(0, ROOT)
    sys.path.insert(0, REF)
    import torch
    spec = importlib.util.spec_from_file_location("external.maskrcnn_benchmark.roi_layers._C", os.path.join(ROOT, "integration", "_C.py"))
    shim = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(shim)
    sys.modules["external.maskrcnn_benchmark.roi_layers._C"] = shim
    sys.modules.setdefault("torchvision", types.ModuleType("torchvision"))
    import external.maskrcnn_benchmark.roi_layers as RL               # the REFERENCE's package
    assert RL.__file__.startswith(REF) and RL.roi_align.__self__.__module__ == "external.maskrcnn_benchmark.roi_layers.roi_align"
    import oracle

    rs = np.random.RandomState(5)
    B, C, H, W = 2, 6, 25, 25
    x = rs.randn(B, C, H, W).astype(np.float32)
    rois = np.array([[0, 0, 0, 399, 399], [1, 33.3, 50.1, 180.7, 222.2], [1, 300, 120, 399, 380], [0, 10, 10, 12, 11], [1, -20, 100, 90, 450]], np.float32)
    g = rs.randn(rois.shape[0], C, 7, 7).astype(np.float32)
    # ROIAlign module (roi_align.py:81-100) forward + backward through the reference's _ROIAlign Function
    for sr in (0, 2):
        xt = torch.from_numpy(x).requires_grad_(True)
        y = RL.ROIAlign((7, 7), 1 / 16., sr)(xt, torch.from_numpy(rois))
        assert np.array_equal(y.detach().numpy(), oracle.roi_align_forward(x, rois, (7, 7), 1 / 16., sr))
        y.backward(torch.from_numpy(g))
        ref = oracle.roi_align_backward(g, rois, (7, 7), 1 / 16., sr, x.shape)
        assert np.abs(xt.grad.numpy() - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())
    # ROIPool module (roi_pool.py) forward + backward through _ROIPool
    xt = torch.from_numpy(x).requires_grad_(True)
    y = RL.ROIPool((7, 7), 1 / 16.)(xt, torch.from_numpy(rois))
    out, arg = oracle.roi_pool_forward(x, rois, (7, 7), 1 / 16.)
    assert np.array_equal(y.detach().numpy(), out)
    y.backward(torch.from_numpy(g))
    assert np.abs(xt.grad.numpy() - oracle.roi_pool_backward(g, arg, rois, (7, 7), x.shape)).max() <= 1e-5
    # nms (nms.py:38): int64 CPU tensor of ascending kept indices; fp32 and fp64; empty input
    for n in (1, 34, 150):
        xy = rs.uniform(0, 300, (n, 2))
        boxes = np.concatenate([xy, xy + rs.uniform(10, 150, (n, 2))], 1)
        scores = rs.permutation(n).astype(np.float64) / n
        k32 = RL.nms(torch.from_numpy(boxes.astype(np.float32)), torch.from_numpy(scores.astype(np.float32)), 0.4)
        assert k32.dtype == torch.int64 and k32.device.type == "cpu"
        assert np.array_equal(k32.numpy(), oracle.nms(boxes.astype(np.float32), scores.astype(np.float32), 0.4))
        k64 = RL.nms(torch.from_numpy(boxes), torch.from_numpy(scores), 0.4)
        assert np.array_equal(k64.numpy(), oracle.nms_f64(boxes, scores, 0.4))
    assert RL.nms(torch.zeros(0, 4), torch.zeros(0), 0.4).numel() == 0
    print("Option B: the reference's roi_layers (ROIAlign / ROIPool modules + autograd Functions, nms) over integration/_C.py == oracle")


if __name__ == "__main__":
    main()
