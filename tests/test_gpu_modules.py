"""The product modules on the real gfx950 library against the reference's golden vectors."""
import pytest

from tests import module_cases as MC

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", MC.GPU_CASES)
def test_gpu_module(name, golden):
    getattr(MC, name)("cuda", golden)
