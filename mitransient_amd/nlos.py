"""Helpers to aim the laser of an NLOS scene.  Public names and argument order follow mitransient/nlos.py:5-70;
all three resolve to one world-space point and share ``_aim``."""
from __future__ import annotations

import numpy as np

from .transform import ScalarTransform4f

_UP = (0.0, 1.0, 0.0)


def _aim(wall, laser, point):
    # Re-orient the laser (its position is kept) and tell the capture meter where it lands: the meter uses
    # the laser->wall distance and the illuminated point when account_first_and_last_bounces is off.
    point = np.asarray(point, dtype=np.float64).reshape(3)
    eye = laser.world_transform().translation()
    laser.to_world = ScalarTransform4f().look_at(origin=eye, target=point, up=list(_UP))
    meter = wall.sensor()
    if meter is not None:
        meter.laser_target = point
        meter.laser_bounce_opl = float(np.linalg.norm(point - eye))


def _wall_point(wall, uv):
    return wall.sample_position(0.0, uv, True).p


def focus_emitter_at_relay_wall_3dpoint(target, relay_wall, emitter):
    """Point the laser at the world-space point ``target`` (nlos.py:5-32)."""
    _aim(relay_wall, emitter, target)


def focus_emitter_at_relay_wall_uv(uv, relay_wall, emitter):
    """Point the laser at the relay-wall point with surface coordinates ``uv`` in [0,1]^2 (nlos.py:35-47)."""
    _aim(relay_wall, emitter, _wall_point(relay_wall, uv))


def focus_emitter_at_relay_wall_pixel(pixel, relay_wall, emitter):
    """Point the laser at the relay-wall point seen by film pixel ``pixel`` (nlos.py:50-70).  The divisor is the
    meter's scan resolution, which for confocal captures differs from the 1x1 film."""
    nx, ny = relay_wall.sensor().film_size
    col, row = (pixel.x, pixel.y) if hasattr(pixel, "x") else (pixel[0], pixel[1])
    _aim(relay_wall, emitter, _wall_point(relay_wall, (float(col) / nx, float(row) / ny)))
