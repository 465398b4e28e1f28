"""The C ABI of the real gfx950 library (step_amd/libstep_amd.so) against the oracle, on the GPU."""
import pytest

from tests import kernel_cases as KC

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bk():
    from tests.backends import GpuBackend

    return GpuBackend()


@pytest.mark.parametrize("name", KC.ALL + KC.GPU_ONLY)
def test_gpu(name, bk, golden):
    getattr(KC, name)(bk, golden)
