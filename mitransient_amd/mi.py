"""A thin stand-in for the slice of ``import mitsuba as mi`` that mitransient notebooks use
around the transient_path hot path: ``set_variant``, ``load_dict``, ``render``, ``traverse``,
``ScalarTransform4f``, ``ScalarColor3d`` (README.md:154-164 of the reference).

    import mitransient_amd.mi as mi
    mi.set_variant('llvm_ad_rgb')          # accepted; the arithmetic always runs on the MI355X
    import mitransient_amd as mitr
    scene = mi.load_dict(mitr.cornell_box())
    steady, transient = mi.render(scene, spp=1024)
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Any, Dict

from . import _cabi, plugins
from . import variant as _variant_mod
from .scene import Properties, flatten_scene, film_desc_from
from .sensors import IndependentSampler, PerspectiveSensor
from .transform import ScalarTransform4f
from .tensor import TensorXf

Transform4f = ScalarTransform4f


def variant():
    return _variant_mod.get()


def set_variant(*names):
    """The unpolarized ``*_ad_rgb`` variants, and ``*_ad_mono`` (needed by ``phasor_hdr_film``): monochromatic rendering
    = every colour replaced by its luminance, one output channel."""
    for n in names:
        if (n.endswith("_rgb") or n.endswith("_mono")) and "polarized" not in n and not n.startswith("scalar"):
            _variant_mod.set(n)
            return
    raise ValueError(f"unsupported variant(s) {names}: mitransient_amd implements the *_ad_rgb / *_ad_mono paths only")


is_monochromatic = _variant_mod.is_monochromatic


__version__ = "3.7.0-mitransient_amd"       # the Mitsuba generation whose plugin semantics are mirrored (reference: >=3.6,<3.9)


def ScalarPoint3f(*v):
    return [float(x) for x in (v[0] if len(v) == 1 else v)]


Point3f = ScalarPoint3f
Point2f = ScalarPoint2f = ScalarPoint3f        # plain float lists, any length


class util:                                 # namespace, like mitsuba.util
    @staticmethod
    def convert_to_bitmap(data, uint8_srgb=True):
        """``mi.util.convert_to_bitmap``: array -> displayable image (uint8 sRGB by default)"""
        import numpy as np
        from .vis import to_srgb_uint8
        a = np.array(data)
        return to_srgb_uint8(a) if uint8_srgb else a


def ScalarColor3d(*v):
    return [float(x) for x in (v[0] if len(v) == 1 else v)]


ScalarColor3f = ScalarColor3d


_SHAPE_TYPES = ("rectangle", "cube", "obj", "ply", "sphere", "disk", "cylinder")


def _nested(d, pred_dict, pred_obj):
    """the first nested plugin of a dictionary that matches: keys are arbitrary (the notebooks write 'transient_film',
    'nlos_sensor', ...), values are dictionaries or objects that mi.load_dict returned earlier"""
    for v in d.values():
        if isinstance(v, dict) and pred_dict(v):
            return v
        if not isinstance(v, dict) and pred_obj(v):
            return v
    return None


def _make_sensor(sd, shape_obj=None):
    from . import films as _f  # noqa: F401  (registers the film plugins)
    from .films.transient_hdr_film import TransientHDRFilm
    fd = _nested(sd, lambda v: str(v.get("type", "")).endswith("_film") or v.get("type") == "hdrfilm",
                 lambda v: isinstance(v, TransientHDRFilm))
    if fd is None:
        raise ValueError("sensor: a 'film' is required")
    film = fd if isinstance(fd, TransientHDRFilm) else plugins.create_film(fd["type"], Properties(fd["type"], fd))
    smp = sd.get("sampler", {"type": "independent"})
    if smp.get("type") != "independent":
        raise ValueError(f"failed to instantiate unknown plugin of type \"{smp.get('type')}\" (supported samplers: independent)")
    sampler = IndependentSampler(Properties("independent", smp))
    if sd["type"] == "perspective":
        return PerspectiveSensor(sd, film, sampler)
    if sd["type"] == "nlos_capture_meter":
        from .sensors.nloscapturemeter import NLOSCaptureMeter
        s = NLOSCaptureMeter(Properties("nlos_capture_meter", sd), film, sampler)
        s.dict_, s.shape_ = sd, shape_obj
        return s
    raise ValueError(f"failed to instantiate unknown plugin of type \"{sd['type']}\" "
                     "(supported sensors: perspective, nlos_capture_meter)")


def _load_shape(d):
    """a shape dictionary -> Shape object; a nested nlos_capture_meter becomes its sensor"""
    from .shapes import Shape
    from .sensors.nloscapturemeter import NLOSCaptureMeter
    sh = Shape(d)
    for k, v in d.items():
        if isinstance(v, dict) and v.get("type") == "nlos_capture_meter":
            sh.sensor_ = _make_sensor(v, sh)
            sh.sensor_key = k
        elif isinstance(v, NLOSCaptureMeter):              # a sensor object loaded on its own (1-simple-nlos-scenes.ipynb)
            sh.sensor_, sh.sensor_key = v, k
            v.shape_ = sh
    return sh


class Scene:
    def __init__(self, d: Dict[str, Any], base_dir: str = ".", approximate_materials: bool = False, geometry=None):
        self.approximate_materials = approximate_materials
        self.geometry_ = geometry              # pre-flattened triangles + tables (scene.load_geometry), or None
        from . import integrators as _i, films as _f  # noqa: F401  (registers the plugins)
        from .shapes import Shape
        from .emitters import Projector
        if d.get("type") != "scene":
            raise ValueError("load_dict(): expected a dictionary with 'type': 'scene'")
        self.base_dir = base_dir
        # objects that were loaded on their own (mi.load_dict(shape) / mi.load_dict(projector)) keep their identity,
        # so that mitransient.nlos.focus_emitter_* edits made later are seen by the render
        self.shape_objs_, self.emitters_, self.sensors_ = {}, [], []
        flat = {}
        for k, v in d.items():
            if isinstance(v, Shape):
                self.shape_objs_[k] = v
                flat[k] = v.dict_
            elif isinstance(v, Projector):
                self.emitters_.append(v)
            elif isinstance(v, dict) and v.get("type") == "projector":
                e = Projector(Properties("projector", v))
                e.dict_ = v
                self.emitters_.append(e)
            elif isinstance(v, dict) and v.get("type") in _SHAPE_TYPES:
                self.shape_objs_[k] = _load_shape(v)
                flat[k] = v
            else:
                flat[k] = v
        self.dict_ = flat
        from .integrators.common import TransientADIntegrator
        integ = [v for v in flat.values() if isinstance(v, TransientADIntegrator) or (isinstance(v, dict) and
                 (str(v.get("type", "")).startswith("transient") or v.get("type") in ("path", "direct")))]
        if len(integ) != 1:
            raise ValueError("load_dict(): exactly one integrator is required")
        idict = integ[0]
        self.integrator_ = idict if isinstance(idict, TransientADIntegrator) else \
            plugins.create_integrator(idict["type"], Properties(idict["type"], idict))
        for k, v in flat.items():
            if isinstance(v, dict) and v.get("type") == "perspective":
                self.sensors_.append(_make_sensor(v))
        self.relay_names_ = {}
        for k, sh in self.shape_objs_.items():
            if sh.sensor() is not None:
                self.sensors_.append(sh.sensor())
                self.relay_names_[id(sh.sensor())] = k
        if not self.sensors_:
            raise ValueError("load_dict(): at least one sensor is required")
        self._data = {}
        self._handles = {}
        self._nlos_fp = {}

    def sensors(self):
        return self.sensors_

    def emitters(self):
        return self.emitters_

    def shapes(self):
        return list(self.shape_objs_.values())

    def integrator(self):
        return self.integrator_

    def data(self, sensor=0):
        """Flat float32 arrays for a sensor (what crosses the C-ABI)."""
        if isinstance(sensor, int):
            sensor = self.sensors_[sensor]
        key = id(sensor)
        if key not in self._data:
            self._data[key] = flatten_scene(self.dict_, sensor.film(), sensor.dict_, self.base_dir,
                                            self.relay_names_.get(key), self.approximate_materials, self.geometry_)
        sd = self._data[key]
        sd.film = film_desc_from(sensor.film())
        from .integrators.transientnlospath import TransientNLOSPath
        # NLOS tier (rebuilt from the live objects): a nlos_capture_meter on its relay wall, or — as in the reference's
        # examples/transient-nlos/nlos-z-*.xml — transient_nlos_path behind an ordinary perspective camera
        if key in self.relay_names_ or isinstance(self.integrator_, TransientNLOSPath):
            from .scene import nlos_desc_from
            if len(self.emitters_) != 1:
                raise AssertionError(f"You have defined multiple ({len(self.emitters_)}) emitters in the scene with a "
                                     "NLOS capture meter. You should have only 1.")
            sd.nlos = nlos_desc_from(self.integrator_, sensor, self.emitters_[0], sd.relay_shape)
        return sd

    def gpu_handle(self, ctx, sensor=0):
        if isinstance(sensor, int):
            sensor = self.sensors_[sensor]
        key = (id(sensor), ctx.device_index)
        sd = self.data(sensor)
        if key not in self._handles:
            h = C.c_void_p()
            d = sd.desc()
            ctx.check(ctx.lib.mtr_scene_create(ctx.handle, C.byref(d), C.byref(h)), "mtr_scene_create")
            self._handles[key] = h
        h = self._handles[key]
        fd = sd.film
        ctx.check(ctx.lib.mtr_scene_set_film(h, C.byref(fd)), "mtr_scene_set_film")
        if sd.nlos is not None:
            # TransientNLOSPath.prepare re-derives its tables on every render; here only when the description changed
            # (laser moved by nlos.focus_emitter_*, integrator property, film size): mtr_scene_set_nlos reallocates
            # four device tables and retraces the scanned points, which is not free per pass / per band
            d = sd.desc()
            n = sd.nlos
            n_val = type(n).from_buffer_copy(n)
            n_val.shapes = None                       # the table is compared by value, not by address
            fp = (bytes(n_val), bytes(memoryview(sd.shapes)),
                  int(fd.width), int(fd.height), int(fd.laser_scan_width), int(fd.laser_scan_height))
            if self._nlos_fp.get(key) != fp:
                ctx.check(ctx.lib.mtr_scene_set_nlos(h, d.nlos), "mtr_scene_set_nlos")
                self._nlos_fp[key] = fp
        return h

    def gpu_traits(self, sensor=0):
        """MTR_TRAIT_* bits of the scene on the current device (which specialised kernels its tables select)"""
        from .runtime import get_context
        ctx = get_context()
        t = C.c_uint32(0)
        ctx.check(ctx.lib.mtr_scene_traits(self.gpu_handle(ctx, sensor), C.byref(t)), "mtr_scene_traits")
        return int(t.value)

    def __del__(self):
        try:
            lib = _cabi.load_library()
            for h in self._handles.values():
                lib.mtr_scene_destroy(h)
        except Exception:
            pass


def load_dict(d: Dict[str, Any], base_dir: str = ".", approximate_materials: bool = False):
    """``mi.load_dict``: a scene dictionary -> Scene; a single shape / projector dictionary -> that object
    (tests/integration/test_nlos.py:86-98 builds the relay wall and the laser this way)."""
    t = d.get("type")
    if t == "scene":
        return Scene(d, base_dir, approximate_materials)
    if t in _SHAPE_TYPES:
        return _load_shape(d)
    if t == "projector":
        from .emitters import Projector
        e = Projector(Properties("projector", d))
        e.dict_ = d
        return e
    from . import integrators as _i, films as _f  # noqa: F401  (registers the plugins)
    if str(t).endswith("_film"):                                # stand-alone plugins, as 1-simple-nlos-scenes.ipynb builds them
        return plugins.create_film(t, Properties(t, d))
    if t in ("nlos_capture_meter", "perspective"):
        return _make_sensor(d)
    if str(t).startswith("transient"):
        return plugins.create_integrator(t, Properties(t, d))
    raise ValueError(f"load_dict(): unsupported top-level plugin type \"{t}\"")


def render(scene: Scene, params=None, sensor=0, integrator=None, seed=0, seed_grad=0, spp=0, spp_grad=0):
    """``mi.render``: returns ``(steady (H,W,3), transient (H,W,T,3))`` like the reference's
    TransientADIntegrator.render (common.py:212-213)."""
    integ = integrator or scene.integrator()
    return integ.render(scene, sensor=sensor, seed=seed, spp=spp)


class _Params(dict):
    def __init__(self, objs):
        super().__init__()
        self._objs = objs
        self._dirty = set()

    def __setitem__(self, k, v):
        self._dirty.add(k)
        super().__setitem__(k, v)

    def update(self, *a, **k):
        if a or k:
            return super().update(*a, **k)
        for key in self._dirty:
            obj, attr = self._objs[key]
            cur = getattr(obj, attr)
            try:
                setattr(obj, attr, type(cur)(self[key]))
            except Exception:
                setattr(obj, attr, self[key])
        self._dirty.clear()


def traverse(obj):
    """``mi.traverse`` for the film parameters the reference exports (transient_hdr_film.py:295-308)."""
    objs = {}

    class _CB:
        def __init__(self, prefix, o):
            self.prefix, self.o = prefix, o

        def put(self, name, value, flags=0):
            objs[self.prefix + name] = (self.o, name)

    targets = []
    if isinstance(obj, Scene):
        for i, s in enumerate(obj.sensors()):
            targets.append((f"sensor{'' if i == 0 else i}.film.", s.film()))
    elif hasattr(obj, "film") and callable(obj.film):               # a sensor: its own keys + "film.*"
        targets.append(("", obj))
        targets.append(("film.", obj.film()))
    else:
        targets.append(("", obj))
    for prefix, o in targets:
        o.traverse(_CB(prefix, o))
    p = _Params(objs)
    for k, (o, attr) in objs.items():
        dict.__setitem__(p, k, getattr(o, attr))
    return p


def load_file(path, **kwargs):
    from .xml_loader import load_file as _lf
    return _lf(path, **kwargs)
