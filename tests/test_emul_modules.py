"""The product's Python host layer (step_amd.backbone / heads / roi_layers) driven end to end on the
HOST interpreter build of the kernels, against the reference's golden vectors.  CPU only."""
import pytest

from tests import module_cases as MC
from tests.emul.patch import emulated_kernels


@pytest.mark.parametrize("name", MC.CPU_CASES)
def test_emul_module(name, golden):
    with emulated_kernels():
        getattr(MC, name)("cpu", golden)
