"""``nlos_capture_meter`` (mitransient/sensors/nloscapturemeter.py): a sensor attached to a relay-wall
shape; rays leave ``sensor_origin`` towards the centres of the film's pixels mapped onto the shape's
UV parameterisation (:136-158, :182-202).  The ray arithmetic runs in the HIP kernels (nlos_sensor_ray)."""
from __future__ import annotations

import numpy as np

from ..scene import Properties


class NLOSCaptureMeter:
    def __init__(self, props: Properties, film, sampler):
        self.film_, self.sampler_ = film, sampler
        self.needs_sample_3 = False
        self.account_first_and_last_bounces = props.get("account_first_and_last_bounces", True)   # deprecated, unused (:96-102)
        o = props.get("sensor_origin", [0.0, 0.0, 0.0])
        self.sensor_origin = np.asarray(o, dtype=np.float64).reshape(3)
        self.laser_bounce_opl = 0.0
        self.laser_target = np.zeros(3)
        # nloscapturemeter.py:111-119: a confocal meter scans one point at a time — a 1 x 1 film, the scan resolution
        # in original_film_*, every sensor ray aimed at laser_target (:142)
        self.original_film_width = props.get("original_film_width", None)
        self.original_film_height = props.get("original_film_height", None)
        if self.original_film_width is None or self.original_film_height is None:
            self.film_size = (float(film.size()[0]), float(film.size()[1]))
            self.is_confocal = False
        else:
            self.film_size = (float(self.original_film_width), float(self.original_film_height))
            self.is_confocal = True
            if tuple(film.size()) != (1, 1):
                raise RuntimeError(f"Confocal configuration requires a film with size [1,1] instead of {list(film.size())}")
        self.shape_ = None
        self.dict_ = None

    def film(self):
        return self.film_

    def sampler(self):
        return self.sampler_

    def shape(self):
        return self.shape_

    get_shape = shape

    def traverse(self, callback):
        for k in ("needs_sample_3", "is_confocal", "laser_bounce_opl", "laser_target"):
            callback.put(k, getattr(self, k), 0)

    def to_string(self):
        return (f"NLOSCaptureMeter[\n  laser_bounce_opl = {self.laser_bounce_opl}, \n"
                f"  is_confocal = {self.is_confocal}, \n  film = {self.film_}, \n]")

    __str__ = __repr__ = to_string
