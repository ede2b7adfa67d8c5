"""Per-device context of the HIP library (one ``mtr_ctx`` per GPU, bound to torch's stream).

PyTorch is plumbing here: it owns device memory (film tensors) and the stream;
every arithmetic step of the path runs in ``libmitransient_amd.so``.
"""
from __future__ import annotations

import ctypes as C

from . import _cabi

_contexts = {}


def _torch():
    import torch
    return torch


def require_gpu():
    torch = _torch()
    if not torch.cuda.is_available():
        raise _cabi.MitransientAMDError(
            "no HIP device visible: mitransient_amd has no CPU path "
            "(the CPU oracle under oracle/ is test infrastructure only)")
    return torch


class Context:
    def __init__(self, device_index: int):
        self.lib = _cabi.load_library()
        self.device_index = device_index
        h = C.c_void_p()
        _cabi.check(self.lib.mtr_ctx_create(device_index, C.byref(h)), None, "mtr_ctx_create")
        self.handle = h

    def bind_current_stream(self):
        torch = _torch()
        s = torch.cuda.current_stream(self.device_index)
        _cabi.check(self.lib.mtr_ctx_set_stream(self.handle, C.c_void_p(s.cuda_stream)), self.handle,
                    "mtr_ctx_set_stream")

    def check(self, status, what=""):
        _cabi.check(status, self.handle, what)


def get_context(device_index=None) -> Context:
    torch = require_gpu()
    if device_index is None:
        device_index = torch.cuda.current_device()
    if device_index not in _contexts:
        _contexts[device_index] = Context(device_index)
    return _contexts[device_index]
