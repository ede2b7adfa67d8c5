"""Host-side 4x4 transforms with the chaining semantics of ``mi.ScalarTransform4f``.

The reference builds its scenes with ``T().translate(..).rotate(..).scale(..)``
(mitransient/utils.py:82,99-103,149) — each call right-multiplies, so the chain
is the matrix product ``A @ B @ C`` (right-most applied first).  All host math
is float64; values are rounded to float32 once, when the scene is flattened.
"""
from __future__ import annotations

import math
import numpy as np


def _v3(v):
    a = np.asarray(v, dtype=np.float64).reshape(-1)
    if a.size == 1:
        a = np.repeat(a, 3)
    if a.size != 3:
        raise ValueError(f"expected a 3-vector, got {v!r}")
    return a


class ScalarTransform4f:
    __slots__ = ("matrix",)

    def __init__(self, matrix=None):
        if matrix is None:
            self.matrix = np.eye(4, dtype=np.float64)
        elif isinstance(matrix, ScalarTransform4f):
            self.matrix = matrix.matrix.copy()
        else:
            m = np.asarray(matrix, dtype=np.float64)
            if m.shape != (4, 4):
                raise ValueError("ScalarTransform4f expects a 4x4 matrix")
            self.matrix = m.copy()

    # -- chaining API ---------------------------------------------------
    def __matmul__(self, other):
        if isinstance(other, ScalarTransform4f):
            return ScalarTransform4f(self.matrix @ other.matrix)
        v = np.asarray(other, dtype=np.float64).reshape(-1)
        if v.size == 3:  # treated as a point, like mi.Transform4f @ Point3f
            return self.transform_affine(v)
        raise TypeError("unsupported operand for @")

    def translate(self, v):
        m = np.eye(4)
        m[:3, 3] = _v3(v)
        return ScalarTransform4f(self.matrix @ m)

    def scale(self, v):
        m = np.eye(4)
        m[0, 0], m[1, 1], m[2, 2] = _v3(v)
        return ScalarTransform4f(self.matrix @ m)

    def rotate(self, axis, angle):
        """Rotation by ``angle`` DEGREES about ``axis`` (Mitsuba convention)."""
        a = _v3(axis)
        a = a / np.linalg.norm(a)
        rad = math.radians(float(angle))
        s, c = math.sin(rad), math.cos(rad)
        x, y, z = a
        m = np.eye(4)
        m[:3, :3] = [
            [x * x + (1 - x * x) * c, x * y * (1 - c) - z * s, x * z * (1 - c) + y * s],
            [x * y * (1 - c) + z * s, y * y + (1 - y * y) * c, y * z * (1 - c) - x * s],
            [x * z * (1 - c) - y * s, y * z * (1 - c) + x * s, z * z + (1 - z * z) * c],
        ]
        return ScalarTransform4f(self.matrix @ m)

    def look_at(self, origin, target, up):
        o, t, u = _v3(origin), _v3(target), _v3(up)
        d = t - o
        d = d / np.linalg.norm(d)
        left = np.cross(u, d)
        n = np.linalg.norm(left)
        if n == 0:
            raise ValueError("look_at(): 'up' is parallel to the viewing direction")
        left = left / n
        new_up = np.cross(d, left)
        m = np.eye(4)
        m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = left, new_up, d, o
        return ScalarTransform4f(self.matrix @ m)

    @staticmethod
    def perspective(fov, near, far):
        """Mitsuba's ``Transform4f::perspective`` (fov in degrees, along x)."""
        recip = 1.0 / (far - near)
        cot = 1.0 / math.tan(math.radians(fov * 0.5))
        m = np.zeros((4, 4))
        m[0, 0] = cot
        m[1, 1] = cot
        m[2, 2] = far * recip
        m[2, 3] = -near * far * recip
        m[3, 2] = 1.0
        return ScalarTransform4f(m)

    def inverse(self):
        return ScalarTransform4f(np.linalg.inv(self.matrix))

    # -- application ------------------------------------------------------
    def transform_affine(self, p):
        p = np.asarray(p, dtype=np.float64)
        r = p @ self.matrix[:3, :3].T + self.matrix[:3, 3]
        return r

    def transform_vector(self, v):
        return np.asarray(v, dtype=np.float64) @ self.matrix[:3, :3].T

    def translation(self):
        return self.matrix[:3, 3].copy()

    def __repr__(self):
        return f"ScalarTransform4f(\n{self.matrix}\n)"


def to_transform(x) -> ScalarTransform4f:
    if x is None:
        return ScalarTransform4f()
    if isinstance(x, ScalarTransform4f):
        return x
    return ScalarTransform4f(np.asarray(x, dtype=np.float64))
