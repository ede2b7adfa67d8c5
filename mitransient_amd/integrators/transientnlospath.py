"""``transient_nlos_path`` plugin (mitransient/integrators/transientnlospath.py): the transient path tracer
with the NLOS-specific sampling routines of [Royo2022] — laser sampling (:511-635), hidden-geometry
sampling (:637-670) — and the Single / Confocal / Exhaustive capture types (Exhaustive with the 6-D
``exhaustive_scan`` film).  Same properties and defaults as the reference (:200-249); the loop (:672-927) runs
in the HIP kernels (``nlos_bounce`` in csrc/mtr_nlos.h)."""
from __future__ import annotations

import enum

from .. import _cabi
from .common import TransientADIntegrator


class CaptureType(enum.IntEnum):                  # transientnlospath.py:12-13
    Single = 1
    Confocal = 2
    Exhaustive = 3


class TransientNLOSPath(TransientADIntegrator):
    def __init__(self, props):
        super().__init__(props)
        self.filter_depth = int(props.get("filter_depth", -1))
        self.filter_bounces = int(props.get("filter_bounces", -1))
        if self.filter_depth != -1 and self.filter_bounces != -1:
            raise AssertionError("Only use one of filter_depth or filter_bounces")
        if self.filter_bounces != -1:
            self.filter_depth = self.filter_bounces + 1
        self.discard_direct_paths = bool(props.get("discard_direct_paths", False))
        ct = props.get("capture_type", 1)
        if isinstance(ct, str):
            ct = {"single": 1, "confocal": 2, "exhaustive": 3}[ct.lower()]
        self.capture_type = int(ct)
        if self.capture_type not in (1, 2, 3):
            raise AssertionError("capture_type must be either an int, a string or a CaptureType enum")
        self.force_equal_grids = bool(props.get("force_equal_illumination_scanning", True))
        self.illumination_scan_fov = float(props.get("illumination_scan_fov", 20.0))
        self.laser_sampling = bool(props.get("nlos_laser_sampling", False))
        self.hg_sampling = bool(props.get("nlos_hidden_geometry_sampling", False))
        self.hg_sampling_do_rroulette = bool(props.get("nlos_hidden_geometry_sampling_do_rroulette", False)) and self.hg_sampling
        self.hg_sampling_includes_relay_wall = (bool(props.get("nlos_hidden_geometry_sampling_includes_relay_wall", False))
                                                and self.hg_sampling)
        self.account_first_and_last_bounces = bool(props.get("account_first_and_last_bounces", False))
        if self.camera_unwarp:
            raise AssertionError("Do not use camera_unwarp with TransientNLOSPath. "
                                 "Use account_first_and_last_bounces instead for the same purpose.")

    def nlos_flags(self):
        f = 0
        f |= _cabi.MTR_NLOS_LASER_SAMPLING if self.laser_sampling else 0
        f |= _cabi.MTR_NLOS_HG_SAMPLING if self.hg_sampling else 0
        f |= _cabi.MTR_NLOS_HG_RROULETTE if self.hg_sampling_do_rroulette else 0
        f |= _cabi.MTR_NLOS_HG_INCLUDES_WALL if self.hg_sampling_includes_relay_wall else 0
        f |= _cabi.MTR_NLOS_ACCOUNT_FIRST_LAST if self.account_first_and_last_bounces else 0
        f |= _cabi.MTR_NLOS_DISCARD_DIRECT if self.discard_direct_paths else 0
        f |= _cabi.MTR_NLOS_FORCE_EQUAL_GRIDS if self.force_equal_grids else 0
        return f

    def check_transient_(self, scene, sensor):
        super().check_transient_(scene, sensor)
        from ..sensors.nloscapturemeter import NLOSCaptureMeter
        if isinstance(sensor, int):
            sensor = scene.sensors()[sensor]
        from ..sensors import PerspectiveSensor
        if not isinstance(sensor, (NLOSCaptureMeter, PerspectiveSensor)):
            raise AssertionError("transient_nlos_path needs a nlos_capture_meter or a perspective sensor")
        film = sensor.film()
        if self.capture_type == 3:
            if not getattr(film, "exhaustive_scan", False):
                raise AssertionError("capture_type 'exhaustive' needs a film with exhaustive_scan=True and "
                                     "laser_scan_width / laser_scan_height")
            if self.force_equal_grids and (film.laser_scan_width, film.laser_scan_height) != tuple(film.size()):
                raise AssertionError("Sensor and laser scan resolution must be equal if "
                                     "force_equal_illumination_scanning is set to True")       # transientnlospath.py:343-345
        if len(scene.emitters()) != 1:
            raise AssertionError(f"You have defined multiple ({len(scene.emitters())}) emitters in the scene with a "
                                 "NLOS capture meter. You should have only 1.")

    def sample(self, *args, **kwargs):
        raise NotImplementedError("TransientNLOSPath.sample() runs inside mtr_render (HIP); use render().")


def register():
    from ..plugins import register_integrator
    register_integrator("transient_nlos_path", lambda props: TransientNLOSPath(props))
