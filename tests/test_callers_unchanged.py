"""The reference's own call sites over the drop-in (build container only: skipped where /root/reference is absent).
oracle/check_callers.py runs the reference's `utils.utils.inference` -- unchanged -- with `dropin/` ahead of the reference on
sys.path (so `models` / `external.maskrcnn_benchmark.roi_layers` resolve to step_amd), kernels on the host interpreter, and
compares its history with step_amd.driver.inference.  Runs in a subprocess: the check re-routes `models` / `utils` / `external`
on sys.path, which must not leak into the other tests."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("STEP_REFERENCE", "/root/reference")


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "utils")), reason="reference tree not available")
@pytest.mark.timeout(900)
def test_reference_inference_runs_unchanged_over_the_dropin():
    r = subprocess.run([sys.executable, "-m", "oracle.check_callers"], cwd=ROOT, capture_output=True, text=True, timeout=850)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "== step_amd.driver.inference" in r.stdout
    assert "DataParallel(base_net)" in r.stdout          # the wrapper lines of test.py:62-98 ran over the drop-in too


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "external")), reason="reference tree not available")
@pytest.mark.timeout(600)
def test_option_b_reference_roi_layers_run_over_the_ctypes_shim():
    """INTEGRATION.md Option B: the reference's own roi_layers Python (autograd Functions, modules, nms) imported unchanged, with
    integration/_C.py in the place of its pybind extension (oracle/check_option_b.py)."""
    r = subprocess.run([sys.executable, "-m", "oracle.check_option_b"], cwd=ROOT, capture_output=True, text=True, timeout=550)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "over integration/_C.py == oracle" in r.stdout
