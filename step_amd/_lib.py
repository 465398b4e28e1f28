"""step_amd/_lib.py -- loads libstep_amd.so (the gfx950 HIP library, include/step_amd.h).

There is NO fallback: if the library is missing, cannot be loaded, or a tensor is not on a ROCm
device, this module raises.  (The oracle under oracle/ and the interpreter under tests/emul are test
infrastructure and are never imported from here.)
"""
import ctypes
import os

from . import _capi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libstep_amd.so")
_LIB = None


def build(verbose=False):
    """Compile step_amd/csrc/*.hip for gfx950 into step_amd/libstep_amd.so (in-tree)."""
    import subprocess

    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j8"]
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.check_call(cmd)
    return LIB_PATH


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "step_amd: %s not found -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C step_amd/csrc`; there is no CPU fallback" % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        _capi.declare(L)
        if L.step_abi_version() != _capi.ABI_VERSION:
            raise RuntimeError("step_amd: ABI mismatch between libstep_amd.so and step_amd/_capi.py")
        _LIB = L
    return _LIB


def stream_ptr(device=None):
    import torch

    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def dptr(t):
    """Raw device address of a torch tensor (None -> NULL).  Refuses non-device tensors."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("step_amd: expected a tensor on a ROCm device, got a %s tensor (no CPU fallback)" % t.device)
    return ctypes.c_void_p(t.data_ptr())
