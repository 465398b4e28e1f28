"""Kernel SOURCE (step_amd/csrc/*.hip) compiled for the host and run on the fiber SIMT interpreter
(tests/emul) against the oracle: validates index logic / tiling / MFMA fragment mapping / arithmetic
order on the CPU-only build container.  The SAME cases run on the real gfx950 build in
tests/test_gpu_kernels.py."""
import pytest

from tests import kernel_cases as KC
from tests.backends import EmuBackend


@pytest.fixture(scope="module")
def bk():
    return EmuBackend()


@pytest.mark.parametrize("name", KC.ALL)
def test_emul(name, bk, golden):
    getattr(KC, name)(bk, golden)
