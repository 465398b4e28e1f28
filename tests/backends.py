"""Two ways to drive the C ABI from the kernel tests:
   EmuBackend -- host SIMT interpreter build (tests/emul), numpy buffers       [CPU tests]
   GpuBackend -- the real libstep_amd.so, torch device buffers                  [pytest -m gpu]
Both expose .lib (declared ctypes library), .dev(np)->buffer with .ptr/.get(), .stream."""
import ctypes

import numpy as np


class _NpBuf:
    def __init__(self, a):
        self.a = None if a is None else np.ascontiguousarray(a)

    @property
    def ptr(self):
        return None if self.a is None else self.a.ctypes.data

    def get(self):
        return self.a


class EmuBackend:
    name = "emul"
    stream = None

    def __init__(self):
        from tests.emul import emul_lib

        self.lib = emul_lib.lib()

    def dev(self, a):
        return _NpBuf(a)


class _TorchBuf:
    def __init__(self, a):
        import torch

        self.dtype = None if a is None else a.dtype
        if a is None:
            self.t = None
        else:
            a = np.ascontiguousarray(a)
            self.t = torch.from_numpy(a.view(np.int16) if a.dtype == np.uint16 else a).cuda()

    @property
    def ptr(self):
        return None if self.t is None else ctypes.c_void_p(self.t.data_ptr())

    def get(self):
        import torch

        torch.cuda.synchronize()
        a = self.t.cpu().numpy()
        return a.view(np.uint16) if self.dtype == np.uint16 else a


class GpuBackend:
    name = "gfx950"

    def __init__(self):
        import torch
        from step_amd import _lib

        assert torch.cuda.is_available()
        self.lib = _lib.lib()
        self.stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def dev(self, a):
        return _TorchBuf(a)
