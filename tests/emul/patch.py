"""tests/emul/patch.py -- TEST-ONLY: run the product's Python host layer (step_amd.ops / backbone /
heads / driver) against the HOST interpreter build of the kernels, on CPU tensors.  This is a
monkeypatch applied from the test side; the product has no switch for it and never imports this."""
import contextlib
import ctypes

from step_amd import _lib


@contextlib.contextmanager
def emulated_kernels():
    from tests.emul import emul_lib

    saved = (_lib._LIB, _lib.dptr, _lib.stream_ptr)
    _lib._LIB = emul_lib.lib()
    _lib.dptr = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
    _lib.stream_ptr = lambda device=None: None
    try:
        yield
    finally:
        _lib._LIB, _lib.dptr, _lib.stream_ptr = saved
