"""``TensorXf``: the array type the plugin surface returns.

The reference returns ``mi.TensorXf`` (common.py:212-213), which notebooks feed
to ``np.array(...)``.  Ours wraps a torch tensor living in HBM and converts on
demand, so ``np.array(transient)`` / ``transient.shape`` / ``transient.numpy()``
keep working.
"""
from __future__ import annotations

import numpy as np


class TensorXf:
    __slots__ = ("_t",)

    def __init__(self, t):
        self._t = t

    @property
    def shape(self):
        return tuple(self._t.shape)

    @property
    def array(self):
        """Flat view, like ``mi.TensorXf.array``."""
        return self._t.reshape(-1)

    def torch(self):
        return self._t

    def numpy(self):
        return self._t.detach().cpu().numpy()

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a.astype(dtype) if dtype is not None else a

    def __getitem__(self, k):
        return TensorXf(self._t[k])

    def __len__(self):
        return self._t.shape[0]

    def __repr__(self):
        return f"TensorXf(shape={self.shape}, device={self._t.device})"
