"""The C restatement against the reference's OWN compiled C++ (oracle/_ref/_C.so) on seeded random
inputs, bit for bit.  Skipped where the reference build is absent.  CPU only; nothing here reads
/root/reference at run time (the .so is prebuilt by `make -C oracle ref`)."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.skipif(not oracle.have_reference_C(), reason="oracle/_ref/_C.so not built")


@pytest.fixture(scope="module")
def refC():
    return oracle.load_reference_C()


def test_exports(refC):
    for name in ("nms", "roi_align_forward", "roi_align_backward", "roi_pool_forward", "roi_pool_backward"):
        assert hasattr(refC, name)          # csrc/vision.cpp:30-36


@pytest.mark.parametrize("seed", range(6))
def test_roi_align_forward_random(refC, seed):
    rs = np.random.RandomState(seed)
    B, C, H, W = rs.randint(1, 4), rs.randint(1, 9), rs.randint(5, 30), rs.randint(5, 30)
    x = rs.randn(B, C, H, W).astype(np.float32)
    K = rs.randint(1, 20)
    xy = rs.uniform(-40, 16 * max(H, W), (K, 2))
    wh = rs.uniform(-10, 16 * max(H, W), (K, 2))
    rois = np.concatenate([rs.randint(0, B, (K, 1)), xy, xy + wh], 1).astype(np.float32)
    ph, pw = rs.randint(1, 9), rs.randint(1, 9)
    for sr in (0, 1, 3):
        ref = refC.roi_align_forward(torch.from_numpy(x), torch.from_numpy(rois), 1 / 16., ph, pw, sr).numpy()
        out = oracle.roi_align_forward(x, rois, (ph, pw), 1 / 16., sr)
        assert np.array_equal(out, ref)


@pytest.mark.parametrize("seed", range(8))
def test_nms_random(refC, seed):
    rs = np.random.RandomState(100 + seed)
    n = int(rs.choice([1, 2, 11, 34, 63, 64, 65, 300]))
    xy = rs.uniform(0, 300, (n, 2))
    wh = rs.uniform(1, 150, (n, 2))
    boxes = np.concatenate([xy, xy + wh], 1).astype(np.float32)
    scores = (rs.permutation(n) / n).astype(np.float32)
    for thr in (0.0, 0.3, 0.4, 0.9):
        ref = refC.nms(torch.from_numpy(boxes), torch.from_numpy(scores), thr).numpy()
        assert np.array_equal(oracle.nms(boxes, scores, thr), ref)


def test_nms_nan_scores_sort_first(refC):
    """NaN scores: torch's descending sort (nms_cpu.cpp:48) ranks them first; the restatement follows."""
    rs = np.random.RandomState(7)
    for n in (12, 34, 80):
        xy = rs.uniform(0, 200, (n, 2))
        boxes = np.concatenate([xy, xy + rs.uniform(20, 150, (n, 2))], 1).astype(np.float32)
        scores = (rs.permutation(n) / n).astype(np.float32)
        scores[[3, n // 2]] = np.nan                                 # two NaNs that each overlap other boxes
        ref = refC.nms(torch.from_numpy(boxes), torch.from_numpy(scores), 0.3).numpy()
        assert len(ref) < n
        assert np.array_equal(oracle.nms(boxes, scores, 0.3), ref)


def test_reference_has_no_cpu_backward_or_roipool(refC):
    # csrc/ROIAlign.h:68, csrc/ROIPool.h:47,68 -- this is why those oracles are pinned indirectly
    x = torch.zeros(1, 1, 4, 4)
    r = torch.tensor([[0., 0, 0, 3, 3]])
    with pytest.raises(RuntimeError):
        refC.roi_pool_forward(x, r, 1.0, 2, 2)
    with pytest.raises(RuntimeError):
        refC.roi_align_backward(torch.zeros(1, 1, 2, 2), r, 1.0, 2, 2, 1, 1, 4, 4, 0)


@pytest.mark.parametrize("seed", [0, 1])
def test_nms_fp64_restatement_matches_the_reference_double_kernel(refC, seed):
    """AT_DISPATCH_FLOATING_TYPES (cpu/nms_cpu.cpp:95): double boxes run nms_cpu_kernel<double>; oracle.nms_f64 is bit-identical."""
    rs = np.random.RandomState(100 + seed)
    for n in (1, 11, 34, 200):
        xy = rs.uniform(0, 300, (n, 2))
        boxes = np.concatenate([xy, xy + rs.uniform(5, 150, (n, 2))], 1)
        scores = rs.permutation(n).astype(np.float64) / n
        for thr in (0.3, 0.4, 0.7):
            ref = refC.nms(torch.from_numpy(boxes), torch.from_numpy(scores), thr).numpy()
            assert np.array_equal(oracle.nms_f64(boxes, scores, thr), ref)
