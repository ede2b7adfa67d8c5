"""Thin shape objects: what ``mi.load_dict({'type': 'rectangle', ...})`` returns, so that the NLOS helpers
(``mitransient.nlos.focus_emitter_at_relay_wall_*``, nlos.py:5-70) have something to hold on to."""
from __future__ import annotations

import numpy as np

from .transform import to_transform


class _PositionSample:
    def __init__(self, p):
        self.p = p


class Shape:
    def __init__(self, d, sensor=None):
        self.dict_ = d
        self.sensor_ = sensor

    def sensor(self):
        return self.sensor_

    def is_rectangle(self):
        return self.dict_.get("type") == "rectangle"

    def sample_position(self, time, uv, active=True):
        """[mitsuba3: Rectangle::sample_position] p = to_world * (2u-1, 2v-1, 0)"""
        if not self.is_rectangle():
            raise NotImplementedError("sample_position is provided for the relay-wall rectangle only")
        tw = to_transform(self.dict_.get("to_world"))
        u, v = float(uv[0]), float(uv[1])
        return _PositionSample(tw.transform_affine(np.array([2 * u - 1, 2 * v - 1, 0.0])))
