"""Helpers to aim the laser of an NLOS scene (mitransient/nlos.py:5-70): same names and arguments."""
from __future__ import annotations

import numpy as np

from .transform import ScalarTransform4f


def focus_emitter_at_relay_wall_3dpoint(target, relay_wall, emitter):
    """Point the laser at ``target`` (nlos.py:5-32): to_world = look_at(origin, target, up=(0,1,0))."""
    sensor = relay_wall.sensor()
    target = np.asarray(target, dtype=np.float64).reshape(3)
    origin = emitter.world_transform().translation()
    emitter.to_world = ScalarTransform4f().look_at(origin=origin, target=target, up=[0, 1, 0])
    if sensor is not None:
        sensor.laser_bounce_opl = float(np.linalg.norm(target - origin))
        sensor.laser_target = target


def focus_emitter_at_relay_wall_uv(uv, relay_wall, emitter):
    """(nlos.py:35-47) uv in [0,1]^2 of the relay wall."""
    target = relay_wall.sample_position(0.0, uv, True).p
    return focus_emitter_at_relay_wall_3dpoint(target, relay_wall, emitter)


def focus_emitter_at_relay_wall_pixel(pixel, relay_wall, emitter):
    """(nlos.py:50-70) pixel of the transient_hdr_film -> uv = pixel / film_size."""
    sensor = relay_wall.sensor()
    fs = sensor.film_size
    px, py = (pixel.x, pixel.y) if hasattr(pixel, "x") else (pixel[0], pixel[1])
    return focus_emitter_at_relay_wall_uv((float(px) / fs[0], float(py) / fs[1]), relay_wall, emitter)
