"""Emitter objects that live outside shapes.  Only the ``projector`` of the NLOS tier
(the laser of mitransient's NLOS scenes: tests/integration/test_nlos.py:29-39,
examples/transient-nlos/nlos_Z.xml:31-34) — `area` emitters are part of their shape."""
from __future__ import annotations

import numpy as np

from .scene import Properties, _color3
from .transform import ScalarTransform4f, to_transform


class Projector:
    """mitsuba's `projector` with a constant `irradiance` [mitsuba3: src/emitters/projector.cpp]:
    a pinhole at ``to_world``'s origin projecting along its +z axis inside ``fov`` degrees."""

    def __init__(self, props: Properties):
        self.to_world = to_transform(props.get("to_world", None))
        self.fov = float(props.get("fov", 0.0))
        if "fov" not in props:
            raise ValueError("projector: 'fov' is required")
        irr = props.get("irradiance", 1.0)
        if isinstance(irr, dict) and irr.get("type") not in ("rgb", "uniform", "spectrum"):
            raise ValueError("projector: only a constant rgb irradiance is supported")
        self.irradiance = _color3(irr, "projector.irradiance")
        self.scale = float(props.get("scale", 1.0))
        self.dict_ = None

    def world_transform(self) -> ScalarTransform4f:
        return self.to_world

    def traverse(self, callback):
        callback.put("to_world", self.to_world, 0)
        callback.put("scale", self.scale, 0)

    def to_string(self):
        return f"Projector[\n  fov = {self.fov},\n  origin = {self.to_world.translation()}\n]"

    __str__ = __repr__ = to_string
