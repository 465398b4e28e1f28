"""tests/emul/emul_lib.py -- numpy front-end to the HOST build of the kernels (fiber SIMT
interpreter, tests/emul/hipemu.*).  TEST INFRASTRUCTURE ONLY: lets the kernel index logic be
checked against the oracle on the CPU-only build container.  Never used by step_amd."""
import ctypes
import os
import subprocess

import numpy as np

from step_amd import _capi

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_build", "libstep_amd_emul.so")
_lib = None

NP_DT = {_capi.F32: np.float32, _capi.BF16: np.uint16, _capi.F16: np.float16}


def lib():
    global _lib
    if _lib is None:
        subprocess.check_call(["make", "-s", "-C", _HERE, "-j8"])
        _lib = _capi.declare(ctypes.CDLL(_PATH))
    return _lib


def ptr(a):
    return None if a is None else a.ctypes.data


def to_bf16_bits(x):
    """float32 -> bfloat16 bit pattern (round to nearest even), as uint16"""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)
    return r


def from_bf16_bits(b):
    return (b.astype(np.uint32) << 16).view(np.float32)


def encode(x, dt):
    x = np.ascontiguousarray(x, np.float32)
    if dt == _capi.F32:
        return x
    if dt == _capi.BF16:
        return to_bf16_bits(x)
    return x.astype(np.float16)


def decode(a, dt):
    if dt == _capi.F32:
        return a
    if dt == _capi.BF16:
        return from_bf16_bits(a)
    return a.astype(np.float32)


def quantize(x, dt):
    return decode(encode(x, dt), dt)
