"""Scene dictionaries -> flat arrays (the host half of ``mi.load_dict``).

Only the subset the north-star path needs is accepted: shapes
``rectangle`` / ``cube`` / ``obj`` / ``ply``, BSDFs ``diffuse`` / ``conductor`` /
``dielectric`` / ``twosided``, the ``area`` emitter on a rectangle or a mesh, the
``perspective`` sensor with an ``independent`` sampler and a
``transient_hdr_film``, and the ``transient_path`` integrator.  Anything else
raises, in the words Mitsuba uses for an unknown plugin.

Reference for the keys and defaults: mitransient/utils.py:78-220 (cornell_box),
mitransient/integrators/common.py:22-30, transient_hdr_film.py:114-121.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Any, Dict, List, Optional

import numpy as np

from . import _cabi
from . import variant as _variant
from .transform import ScalarTransform4f, to_transform


class Properties:
    """Minimal stand-in for ``mi.Properties``: ``get(name, default)`` + plugin name."""

    def __init__(self, plugin_name: str = "", values: Optional[Dict[str, Any]] = None):
        self._plugin = plugin_name
        self._v = dict(values or {})
        self._queried = set()

    def plugin_name(self):
        return self._plugin

    def has_property(self, k):
        return k in self._v

    def get(self, k, default=None):
        self._queried.add(k)
        return self._v.get(k, default)

    def __getitem__(self, k):
        self._queried.add(k)
        return self._v[k]

    def __setitem__(self, k, v):
        self._v[k] = v

    def __contains__(self, k):
        return k in self._v

    def keys(self):
        return self._v.keys()

    def unqueried(self):
        return [k for k in self._v if k not in self._queried and k != "type"]


def _srgb_to_linear(a):
    a = np.asarray(a, dtype=np.float64)
    return np.where(a <= 0.04045, a / 12.92, ((a + 0.055) / 1.055) ** 2.4)


def bitmap_mean_colour(path: str, raw: bool = False) -> np.ndarray:
    """mean linear RGB of an 8-bit image (the approximate-materials stand-in for a `bitmap` texture)"""
    from PIL import Image
    with Image.open(path) as im:
        a = np.asarray(im.convert("RGB"), dtype=np.float64) / 255.0
    if not raw:
        a = _srgb_to_linear(a)
    return a.reshape(-1, 3).mean(axis=0)


def load_bitmap_texture(path: str, raw: bool = False, max_size: Optional[int] = None) -> np.ndarray:
    """8-bit image -> (H, W, 3) float32 linear RGB, row 0 first [mitsuba3: BitmapTexture with raw = false decodes sRGB];
    ``max_size``: box-downsample so that neither side exceeds it (data fixtures of large textures)"""
    from PIL import Image
    with Image.open(path) as im:
        im = im.convert("RGB")
        if max_size and max(im.size) > max_size:
            k = max_size / float(max(im.size))
            im = im.resize((max(1, round(im.size[0] * k)), max(1, round(im.size[1] * k))), Image.BOX)
        a = np.asarray(im, dtype=np.float64) / 255.0
    return decode_texture_u8(np.round(a * 255.0).astype(np.uint8), raw)


def decode_texture_u8(u8: np.ndarray, raw: bool = False) -> np.ndarray:
    a = np.asarray(u8, dtype=np.float64) / 255.0
    if not raw:
        a = _srgb_to_linear(a)
    return np.ascontiguousarray(a, dtype=np.float32)


def _color3(v, what="color", approx_base=None):
    if isinstance(v, dict):
        t = v.get("type")
        if t == "bitmap" and approx_base is not None:
            fn = v.get("filename")
            if not os.path.isabs(fn):
                fn = os.path.join(approx_base, fn)
            return bitmap_mean_colour(fn, bool(v.get("raw", False)))
        if t not in ("rgb", "spectrum", "uniform"):
            raise ValueError(f"failed to instantiate unknown plugin of type \"{t}\" ({what}: only rgb/constant values are supported)")
        v = v.get("value")
    a = np.asarray(v, dtype=np.float64).reshape(-1)
    if a.size == 1:
        a = np.repeat(a, 3)
    if a.size != 3:
        raise ValueError(f"{what}: expected a scalar or an RGB triple")
    if _variant.is_monochromatic():              # [mitsuba3: luminance(Color3f)], f32 like the srgb spectrum plugin in mono mode
        c = a.astype(np.float32)
        lum = (c[0] * np.float32(0.212671) + c[1] * np.float32(0.715160)) + c[2] * np.float32(0.072169)
        a = np.repeat(np.float64(lum), 3)
    return a


# complex IOR presets of mitsuba's `conductor` (subset; values = mitsuba's RGB-mode table) [upstream-unverified]
_CONDUCTOR_PRESETS = {
    "none": ((0.0, 0.0, 0.0), (1.0, 1.0, 1.0)),
}

_IOR_PRESETS = {"vacuum": 1.0, "air": 1.000277, "water": 1.3330, "bk7": 1.5046, "diamond": 2.419,
                "acrylic glass": 1.49, "polypropylene": 1.49, "pyrex": 1.470, "fused quartz": 1.458}


def _ior(v, default):
    if v is None:
        v = default
    if isinstance(v, str):
        if v not in _IOR_PRESETS:
            raise ValueError(f"unknown IOR preset '{v}'")
        return float(_IOR_PRESETS[v])
    return float(v)


_EMITTER_TYPES = ("area", "angulararea", "point", "spot", "projector", "constant", "envmap", "directional")


def fresnel_diffuse_reflectance(eta):
    """[mitsuba3: fresnel_diffuse_reflectance] the diffuse Fresnel reflectance of a dielectric boundary, by the two published
    fits mitsuba cherry-picks from: Egan & Hilgeman (1973) for eta < 1, d'Eon & Irving (2011) otherwise (float32 arithmetic)"""
    f = np.float32
    eta = f(eta); inv = f(1.0) / eta
    if eta < f(1.0):
        return f(f(0.0636) * inv + (eta * (eta * f(-1.4399) + f(0.7099)) + f(0.6681)))
    acc = f(-1.36881)
    for c in (4.98554, -7.80989, 6.75335, -3.4793, 0.919317):
        acc = f(acc * inv + f(c))
    return acc


class _SceneBuilder:
    def __init__(self, d: Dict[str, Any], base_dir: str = ".", approximate_materials: bool = False):
        self.d = d
        self.base_dir = base_dir
        self.approx = approximate_materials
        self.mesh_cache: Dict[str, np.ndarray] = {}
        self.tri_verts: List[np.ndarray] = []
        self.tri_mat: List[np.ndarray] = []
        self.tri_em: List[np.ndarray] = []
        self.tri_uv: List[Optional[np.ndarray]] = []          # per shape: (n, 6) corner texture coordinates, or None
        self.tri_normals: List[Optional[np.ndarray]] = []     # per shape: (n, 9) world-space corner normals (smooth shading), or None = flat
        self.textures: List[np.ndarray] = []                  # (H, W, 3) float32 linear RGB bitmaps referenced by materials
        self.texture_cache: Dict[Any, int] = {}
        self.materials: List[_cabi.mtr_material] = []
        self.mat_cache: Dict[int, int] = {}
        self.emitters: List[_cabi.mtr_emitter] = []
        self.shape_names: List[str] = []
        self.shape_ranges: List[tuple] = []
        self.shapes: List[_cabi.mtr_shape] = []

    # -- BSDFs -------------------------------------------------------------
    def _resolve(self, v):
        if isinstance(v, dict) and v.get("type") == "ref":
            rid = v["id"]
            if rid not in self.d:
                raise ValueError(f"reference to unknown object '{rid}'")
            return self.d[rid], ("ref", rid)
        return v, ("obj", id(v))

    def material_index(self, v) -> int:
        bd, key = self._resolve(v)
        if key in self.mat_cache:
            return self.mat_cache[key]
        m = self._make_material(bd)
        self.materials.append(m)
        self.mat_cache[key] = len(self.materials) - 1
        return self.mat_cache[key]

    def _albedo(self, m, v, what, ab):
        """the colour `a` of a material: a constant, or (approximate_materials unset or 'textures') a bitmap texture whose
        mean stands in wherever a single colour is needed"""
        if isinstance(v, dict) and v.get("type") == "bitmap" and self.approx in (False, None, "textures"):
            if v.get("filter_type", "bilinear") != "bilinear" or v.get("wrap_mode", "repeat") != "repeat" or "to_uv" in v:
                raise ValueError(f"{what}: bitmap textures are available with filter_type = bilinear, wrap_mode = repeat and no to_uv")
            fn = v.get("filename")
            if not os.path.isabs(fn):
                fn = os.path.join(self.base_dir, fn)
            key = (fn, bool(v.get("raw", False)))
            if key not in self.texture_cache:
                self.textures.append(load_bitmap_texture(fn, key[1]))
                self.texture_cache[key] = len(self.textures) - 1
            m.albedo_texture = self.texture_cache[key] + 1
            return self.textures[m.albedo_texture - 1].reshape(-1, 3).mean(axis=0).astype(np.float64)
        return _color3(v, what, ab)

    def _make_material(self, bd) -> _cabi.mtr_material:
        m = _cabi.mtr_material()
        m.int_ior, m.ext_ior = 1.0, 1.0
        for k in range(3):
            m.c[k] = 1.0
            m.c2[k] = 1.0
        t = bd.get("type")
        ab = self.base_dir if self.approx and self.approx != "textures" else None      # bitmap -> mean colour only when approximating
        if self.approx:
            # (approximate_materials="textures": bump maps ignored like True, but bitmaps on reflectances stay textures)
            # nearest material of the hot path's model (opt-in, documented in DESIGN.md): bitmap textures become their mean
            # colour, bump/normal maps are ignored, the smooth plastic coat is dropped; with approximate_materials="smooth"
            # (the config-5 bench fixture) the GGX lobes of roughconductor / roughplastic also collapse to their smooth limit
            if t in ("bumpmap", "normalmap"):
                inner = [v for k, v in bd.items() if isinstance(v, dict) and v.get("type") not in ("bitmap", None)
                         and k != "type" and "filename" not in v]
                if len(inner) != 1:
                    raise ValueError(f"{t}: exactly one nested BSDF is expected")
                return self._make_material(self._resolve(inner[0])[0])
            if t == "roughconductor" and self.approx == "smooth":
                bd = dict(bd, type="conductor"); t = "conductor"
            elif t in ("plastic", "roughplastic") and self.approx == "smooth":
                bd = {"type": "diffuse", "reflectance": bd.get("diffuse_reflectance", 0.5)}; t = "diffuse"
            elif t == "roughdielectric" and self.approx == "smooth":
                bd = dict(bd, type="dielectric"); t = "dielectric"
        if t == "twosided":
            inner = [v for k, v in bd.items() if isinstance(v, dict) and k != "type"]
            if len(inner) != 1:
                raise ValueError("twosided: exactly one nested BSDF is supported")
            inner_d, _ = self._resolve(inner[0])
            m = self._make_material(inner_d)
            if m.type in (_cabi.MTR_BSDF_DIELECTRIC, _cabi.MTR_BSDF_ROUGHDIELECTRIC, _cabi.MTR_BSDF_THINDIELECTRIC):
                raise ValueError("twosided: only materials without a transmission component can be nested")
            m.flags |= _cabi.MTR_MAT_TWOSIDED
            return m
        if t == "diffuse":
            m.type = _cabi.MTR_BSDF_DIFFUSE
            refl = self._albedo(m, bd.get("reflectance", 0.5), "diffuse.reflectance", ab)
            for k in range(3):
                m.a[k] = np.float32(refl[k])
        elif t == "conductor":
            m.type = _cabi.MTR_BSDF_CONDUCTOR
            if "material" in bd and bd["material"] != "none":
                raise ValueError("conductor: material presets other than 'none' are not available; pass eta/k")
            eta = _color3(bd.get("eta", 0.0), "conductor.eta")
            kk = _color3(bd.get("k", 1.0), "conductor.k")
            sr = _color3(bd.get("specular_reflectance", 1.0), "conductor.specular_reflectance")
            for k in range(3):
                m.a[k], m.b[k], m.c[k] = np.float32(eta[k]), np.float32(kk[k]), np.float32(sr[k])
        elif t == "dielectric":
            m.type = _cabi.MTR_BSDF_DIELECTRIC
            m.int_ior = np.float32(_ior(bd.get("int_ior"), "bk7"))
            m.ext_ior = np.float32(_ior(bd.get("ext_ior"), "air"))
            sr = _color3(bd.get("specular_reflectance", 1.0), "dielectric.specular_reflectance")
            st = _color3(bd.get("specular_transmittance", 1.0), "dielectric.specular_transmittance")
            for k in range(3):
                m.c[k], m.c2[k] = np.float32(sr[k]), np.float32(st[k])
        elif t == "thindielectric":
            # [mitsuba3: src/bsdfs/thindielectric.cpp] a thin slab: reflection and straight-through transmission, both delta lobes
            m.type = _cabi.MTR_BSDF_THINDIELECTRIC
            m.int_ior = np.float32(_ior(bd.get("int_ior"), "bk7"))
            m.ext_ior = np.float32(_ior(bd.get("ext_ior"), "air"))
            sr = _color3(bd.get("specular_reflectance", 1.0), "thindielectric.specular_reflectance")
            st = _color3(bd.get("specular_transmittance", 1.0), "thindielectric.specular_transmittance")
            for k in range(3):
                m.c[k], m.c2[k] = np.float32(sr[k]), np.float32(st[k])
        elif t == "plastic":
            # [mitsuba3: src/bsdfs/plastic.cpp] smooth dielectric coat over a diffuse base with internal scattering
            m.type = _cabi.MTR_BSDF_PLASTIC
            m.int_ior = np.float32(_ior(bd.get("int_ior"), "polypropylene"))
            m.ext_ior = np.float32(_ior(bd.get("ext_ior"), "air"))
            diff = self._albedo(m, bd.get("diffuse_reflectance", 0.5), "plastic.diffuse_reflectance", ab)
            sr = _color3(bd.get("specular_reflectance", 1.0), "plastic.specular_reflectance", ab)
            for k in range(3):
                m.a[k], m.c[k] = np.float32(diff[k]), np.float32(sr[k])
            if bd.get("nonlinear", False):
                m.flags |= _cabi.MTR_MAT_NONLINEAR
            eta = np.float32(m.int_ior) / np.float32(m.ext_ior)
            m.internal_reflectance = fresnel_diffuse_reflectance(np.float32(1.0) / eta)        # m_fdr_int
            d_mean = float(np.mean([np.float32(x) for x in diff])); s_mean = float(np.mean([np.float32(x) for x in sr]))
            m.specular_sampling_weight = np.float32(s_mean / (d_mean + s_mean))
        elif t == "roughdielectric":
            # [mitsuba3: src/bsdfs/roughdielectric.cpp] rough refractive interface: distribution (beckmann by default), alpha or
            # alpha_u + alpha_v, int_ior / ext_ior, specular_reflectance / specular_transmittance; visible-normal sampling
            m.type = _cabi.MTR_BSDF_ROUGHDIELECTRIC
            distribution = str(bd.get("distribution", "beckmann"))
            if distribution not in ("ggx", "beckmann"):
                raise ValueError(f"roughdielectric: distribution must be \"beckmann\" or \"ggx\", not \"{distribution}\"")
            if distribution == "beckmann":
                m.flags |= _cabi.MTR_MAT_BECKMANN
            if not bd.get("sample_visible", True):
                raise ValueError("roughdielectric: sample_visible = false is not available")
            alpha = bd.get("alpha", 0.1)
            if "alpha_u" in bd or "alpha_v" in bd:
                if "alpha" in bd or not ("alpha_u" in bd and "alpha_v" in bd):
                    raise ValueError("roughdielectric: specify either alpha or alpha_u and alpha_v")
                alpha = bd["alpha_u"]
                if isinstance(bd["alpha_v"], dict):
                    raise ValueError("roughdielectric: textured alpha is not available")
                if not isinstance(alpha, dict) and np.float32(bd["alpha_v"]) != np.float32(alpha):
                    m.flags |= _cabi.MTR_MAT_ANISOTROPIC
                    m.b[0] = np.float32(bd["alpha_v"])           # (c2 is the transmittance: alpha_v travels in b[0])
            if isinstance(alpha, dict):
                raise ValueError("roughdielectric: textured alpha is not available")
            m.alpha = np.float32(alpha)
            m.int_ior = np.float32(_ior(bd.get("int_ior"), "bk7"))
            m.ext_ior = np.float32(_ior(bd.get("ext_ior"), "air"))
            if m.int_ior == m.ext_ior:
                raise ValueError("roughdielectric: the interior and exterior indices of refraction must differ")
            sr = _color3(bd.get("specular_reflectance", 1.0), "roughdielectric.specular_reflectance")
            st = _color3(bd.get("specular_transmittance", 1.0), "roughdielectric.specular_transmittance")
            for k in range(3):
                m.c[k], m.c2[k] = np.float32(sr[k]), np.float32(st[k])
        elif t in ("roughconductor", "roughplastic"):
            # microfacet lobes [mitsuba3: src/bsdfs/roughconductor.cpp, roughplastic.cpp]: isotropic alpha, visible-normal sampling,
            # distribution = beckmann (mitsuba's default) | ggx
            distribution = str(bd.get("distribution", "beckmann"))
            if distribution not in ("ggx", "beckmann"):
                raise ValueError(f"{t}: distribution must be \"beckmann\" or \"ggx\", not \"{distribution}\"")
            if distribution == "beckmann":
                m.flags |= _cabi.MTR_MAT_BECKMANN
            if not bd.get("sample_visible", True):
                raise ValueError(f"{t}: sample_visible = false is not available")
            alpha = bd.get("alpha", 0.1)
            if "alpha_u" in bd or "alpha_v" in bd:
                # [roughconductor.cpp] either `alpha` or BOTH `alpha_u` and `alpha_v`; roughplastic has a single alpha
                if t != "roughconductor":
                    raise ValueError(f"{t}: alpha_u / alpha_v are parameters of roughconductor; roughplastic takes `alpha`")
                if "alpha" in bd or not ("alpha_u" in bd and "alpha_v" in bd):
                    raise ValueError("roughconductor: specify either alpha or alpha_u and alpha_v")
                alpha = bd["alpha_u"]
                if isinstance(bd["alpha_v"], dict):
                    raise ValueError(f"{t}: textured alpha is not available")
                if np.float32(bd["alpha_v"]) != np.float32(alpha) if not isinstance(alpha, dict) else False:
                    m.flags |= _cabi.MTR_MAT_ANISOTROPIC
                    m.c2[0] = np.float32(bd["alpha_v"])
            if isinstance(alpha, dict):
                raise ValueError(f"{t}: textured alpha is not available")
            m.alpha = np.float32(alpha)
            sr = _color3(bd.get("specular_reflectance", 1.0), f"{t}.specular_reflectance", ab)
            if t == "roughconductor":
                m.type = _cabi.MTR_BSDF_ROUGHCONDUCTOR
                if "material" in bd and bd["material"] != "none":
                    raise ValueError("roughconductor: material presets other than 'none' are not available; pass eta/k")
                eta = _color3(bd.get("eta", 0.0), "roughconductor.eta")
                kk = _color3(bd.get("k", 1.0), "roughconductor.k")
                for k in range(3):
                    m.a[k], m.b[k], m.c[k] = np.float32(eta[k]), np.float32(kk[k]), np.float32(sr[k])
            else:
                from .microfacet import rough_plastic_tables
                m.type = _cabi.MTR_BSDF_ROUGHPLASTIC
                m.int_ior = np.float32(_ior(bd.get("int_ior"), "polypropylene"))
                m.ext_ior = np.float32(_ior(bd.get("ext_ior"), "air"))
                diff = self._albedo(m, bd.get("diffuse_reflectance", 0.5), "roughplastic.diffuse_reflectance", ab)
                for k in range(3):
                    m.a[k], m.c[k] = np.float32(diff[k]), np.float32(sr[k])
                if bd.get("nonlinear", False):
                    m.flags |= _cabi.MTR_MAT_NONLINEAR
                eta = float(np.float32(m.int_ior) / np.float32(m.ext_ior))
                ext, internal = rough_plastic_tables(float(m.alpha), eta, distribution)
                for k in range(_cabi.MTR_ROUGH_TRANSMITTANCE_RES):
                    m.external_transmittance[k] = ext[k]
                m.internal_reflectance = internal
                d_mean = float(np.mean([np.float32(x) for x in diff])); s_mean = float(np.mean([np.float32(x) for x in sr]))
                m.specular_sampling_weight = np.float32(s_mean / (d_mean + s_mean))
        else:
            raise ValueError(f"failed to instantiate unknown plugin of type \"{t}\" (supported BSDFs: diffuse, conductor, dielectric, thindielectric, plastic, roughconductor, roughplastic, roughdielectric, twosided)")
        return m

    # -- shapes ------------------------------------------------------------
    def add_shape(self, name: str, sd: Dict[str, Any]):
        t = sd.get("type")
        tw = to_transform(sd.get("to_world"))
        uv = None
        normals = None
        if t == "rectangle":
            # an analytic primitive (mtr_shape.is_rectangle); the two triangles carry its material / emitter / index
            corners = np.array([[-1, -1, 0], [1, -1, 0], [1, 1, 0], [-1, 1, 0]], dtype=np.float64)
            w = tw.transform_affine(corners)
            tris = np.stack([w[[0, 1, 2]], w[[0, 2, 3]]])
        elif t == "cube":
            ct, uv = _cube_tris()
            tris = tw.transform_affine(ct.reshape(-1, 3)).reshape(-1, 3, 3)
        elif t in ("obj", "ply"):
            fn = sd.get("filename")
            if not os.path.isabs(fn):
                fn = os.path.join(self.base_dir, fn)
            if fn not in self.mesh_cache:
                self.mesh_cache[fn] = load_obj(fn, with_uv=True, with_normals=True) if t == "obj" else load_ply(fn, with_attributes=True)
            v, uv, vn = self.mesh_cache[fn]
            tris = tw.transform_affine(v.reshape(-1, 3)).reshape(-1, 3, 3)
            # shading normals [mitsuba3: Mesh::compute_surface_interaction]: interpolated vertex normals unless face_normals;
            # normals go to world space by the inverse transpose and are renormalised (obj.cpp)
            if vn is not None and not sd.get("face_normals", False):
                nt = vn.reshape(-1, 3) @ np.linalg.inv(tw.matrix[:3, :3])
                ln = np.linalg.norm(nt, axis=1, keepdims=True)
                normals = np.where(ln > 0, nt / np.where(ln > 0, ln, 1.0), 0.0).reshape(-1, 9)
        else:
            raise ValueError(f"failed to instantiate unknown plugin of type \"{t}\" (supported shapes: rectangle, cube, obj, ply)")
        if sd.get("flip_normals", False):
            tris = tris[:, [0, 2, 1], :]
            if uv is not None:
                uv = uv.reshape(-1, 3, 2)[:, [0, 2, 1], :].reshape(-1, 6)
            if normals is not None:
                normals = -normals.reshape(-1, 3, 3)[:, [0, 2, 1], :].reshape(-1, 9)
        bsdf, em = None, None
        for k, v in sd.items():
            # nested plugins are recognised by type, not by key (keys are arbitrary, as in mitsuba): emitter, sensor,
            # anything else is the BSDF
            if not isinstance(v, dict):
                continue
            vt = str(v.get("type", ""))
            if vt in _EMITTER_TYPES:
                em = v
            elif vt.startswith("nlos_") or vt in ("perspective", "irradiancemeter"):
                continue
            else:
                bsdf = v
        if bsdf is None:
            bsdf = {"type": "diffuse", "reflectance": 0.5}     # mitsuba's default BSDF
        mi_ = self.material_index(bsdf)
        em_index = -1
        if em is not None:
            if em.get("type") != "area":
                raise ValueError(f"failed to instantiate unknown plugin of type \"{em.get('type')}\" (supported emitters: area)")
            e = _cabi.mtr_emitter()
            rad = _color3(em.get("radiance", 1.0), "area.radiance")
            if t == "rectangle":
                c = tw.transform_affine(np.zeros(3))
                du = tw.transform_affine(np.array([1.0, 0, 0])) - c
                dv = tw.transform_affine(np.array([0, 1.0, 0])) - c
                for k in range(3):
                    e.center[k], e.du[k], e.dv[k] = np.float32(c[k]), np.float32(du[k]), np.float32(dv[k])
                e.flip_normals = 1 if sd.get("flip_normals", False) else 0      # [Rectangle::sample_position: ps.n = -frame.n]
            else:                                   # triangle-mesh emitter: sampled by face area [Mesh::sample_position]
                e.is_mesh = 1
                e.first_tri = sum(a.shape[0] for a in self.tri_verts)
                e.n_tris = tris.shape[0]
            for k in range(3):
                e.radiance[k] = np.float32(rad[k])
            self.emitters.append(e)
            em_index = len(self.emitters) - 1
        n = tris.shape[0]
        first = sum(a.shape[0] for a in self.tri_verts)
        self.tri_verts.append(tris.astype(np.float32))
        self.tri_mat.append(np.full(n, mi_, dtype=np.uint32))
        self.tri_em.append(np.full(n, em_index, dtype=np.int32))
        self.tri_uv.append(None if uv is None else np.asarray(uv, dtype=np.float32).reshape(n, 6))
        self.tri_normals.append(None if normals is None else np.asarray(normals, dtype=np.float32).reshape(n, 9))
        self.shape_names.append(name)
        self.shape_ranges.append((first, first + n))
        sh = _cabi.mtr_shape()
        sh.first_tri, sh.n_tris, sh.is_rectangle = first, n, 1 if t == "rectangle" else 0
        if t == "rectangle" and sd.get("flip_normals", False):
            # the analytic primitive takes its normal from du x dv, not from the carrier triangles' winding (swapped above):
            # mitsuba's Rectangle negates the frame normal and keeps the parameterisation (MTR_RECT_FLIP_NORMALS)
            sh.is_rectangle |= _cabi.MTR_RECT_FLIP_NORMALS
        sh.has_to_world = 1                       # object -> world of the mesh (an acceleration hint: oriented bounds)
        for i, x in enumerate(np.asarray(tw.matrix, dtype=np.float64)[:3, :].reshape(-1)):
            sh.to_world[i] = np.float32(x)
        if t == "rectangle":
            c = tw.transform_affine(np.zeros(3))
            du = tw.transform_affine(np.array([1.0, 0, 0])) - c
            dv = tw.transform_affine(np.array([0, 1.0, 0])) - c
            for k in range(3):
                sh.center[k], sh.du[k], sh.dv[k] = np.float32(c[k]), np.float32(du[k]), np.float32(dv[k])
        self.shapes.append(sh)


def _cube_tris():
    """[mitsuba3: src/shapes/cube.cpp] (upstream-unverified, from memory of the plugin's tables): [-1,1]^3 as 24 vertices
    (4 per face, with the face's normal and the texture coordinates (0,1) (1,1) (1,0) (0,0)) and 12 triangles
    {0,1,2} {3,0,2} per face, outward-facing.  Returns (triangles (12,3,3), corner texture coordinates (12,6))."""
    verts = np.array([
        [1, -1, -1], [1, -1, 1], [-1, -1, 1], [-1, -1, -1],      # y = -1
        [1, 1, -1], [-1, 1, -1], [-1, 1, 1], [1, 1, 1],          # y = +1
        [1, -1, -1], [1, 1, -1], [1, 1, 1], [1, -1, 1],          # x = +1
        [1, -1, 1], [1, 1, 1], [-1, 1, 1], [-1, -1, 1],          # z = +1
        [-1, -1, 1], [-1, 1, 1], [-1, 1, -1], [-1, -1, -1],      # x = -1
        [1, 1, -1], [1, -1, -1], [-1, -1, -1], [-1, 1, -1]],     # z = -1
        dtype=np.float64)
    uvs = np.tile(np.array([[0, 1], [1, 1], [1, 0], [0, 0]], dtype=np.float32), (6, 1))
    idx = np.array([[4 * f + a, 4 * f + b, 4 * f + c] for f in range(6) for (a, b, c) in ((0, 1, 2), (3, 0, 2))])
    return verts[idx], uvs[idx].reshape(-1, 6)


def load_obj(path: str, with_uv: bool = False, with_normals: bool = False):
    """Wavefront OBJ -> (n,3,3) float64 triangle soup (positions; fan-triangulated; ``l`` and groups are ignored).
    ``with_uv``: also the corner texture coordinates (n,6) float32 — or None when the file has no ``vt`` or some face
    corner lacks one — which only orient the shading frame (mtr_scene_desc.tri_uv).  ``with_normals``: also the corner
    normals (n,3,3) float64 from ``vn`` — or, when the file has none (or a corner lacks one), the angle-weighted vertex
    normals mitsuba computes in that case (Mesh::recompute_vertex_normals over vertices that share position and texture
    index); returns (tris, uv, normals)."""
    verts, uvs, tris, tuv = [], [], [], []
    vns, tvn = [], []
    with open(path, "r") as fh:
        for line in fh:
            if line.startswith("v "):
                p = line.split()
                verts.append((float(p[1]), float(p[2]), float(p[3])))
            elif line.startswith("vt "):
                p = line.split()
                uvs.append((float(p[1]), float(p[2]) if len(p) > 2 else 0.0))
            elif line.startswith("vn "):
                p = line.split()
                vns.append((float(p[1]), float(p[2]), float(p[3])))
            elif line.startswith("f "):
                idx, tix, nix = [], [], []
                for tok in line.split()[1:]:
                    parts = tok.split("/")
                    i = int(parts[0])
                    idx.append(i - 1 if i > 0 else len(verts) + i)
                    if len(parts) > 1 and parts[1]:
                        j = int(parts[1])
                        tix.append(j - 1 if j > 0 else len(uvs) + j)
                    else:
                        tix.append(-1)
                    if len(parts) > 2 and parts[2]:
                        j = int(parts[2])
                        nix.append(j - 1 if j > 0 else len(vns) + j)
                    else:
                        nix.append(-1)
                for k in range(1, len(idx) - 1):
                    tris.append((idx[0], idx[k], idx[k + 1]))
                    tuv.append((tix[0], tix[k], tix[k + 1]))
                    tvn.append((nix[0], nix[k], nix[k + 1]))
    v = np.asarray(verts, dtype=np.float64)
    t = np.asarray(tris, dtype=np.int64).reshape(-1, 3)
    if not with_uv and not with_normals:
        return v[t]
    tu = np.asarray(tuv, dtype=np.int64).reshape(-1, 3)
    uv = None
    if uvs and tu.size and tu.min() >= 0:
        uv = np.asarray(uvs, dtype=np.float32)[tu].reshape(-1, 6).copy()
        uv[:, 1::2] = np.float32(1.0) - uv[:, 1::2]          # [mitsuba3: obj.cpp flip_tex_coords = true] (the frame is unaffected)
    if not with_normals:
        return v[t], uv
    tn = np.asarray(tvn, dtype=np.int64).reshape(-1, 3)
    if vns and tn.size and tn.min() >= 0:
        normals = np.asarray(vns, dtype=np.float64)[tn]
    else:
        normals = vertex_normals(v, t, tu)
    return v[t], uv, normals


def vertex_normals(v: np.ndarray, t: np.ndarray, key2: Optional[np.ndarray] = None) -> np.ndarray:
    """[mitsuba3: Mesh::recompute_vertex_normals] corner normals (n,3,3) of an indexed mesh without normals: every face adds
    its unit normal, weighted by its angle at the corner, to the vertices it touches; a "vertex" is what the loader keeps
    apart: position index (and texture index ``key2`` when given)."""
    P = v[t]                                                   # (n,3,3)
    fn = np.cross(P[:, 1] - P[:, 0], P[:, 2] - P[:, 0])
    ln = np.linalg.norm(fn, axis=1, keepdims=True)
    fn = np.where(ln > 0, fn / np.where(ln > 0, ln, 1.0), 0.0)
    key = t if key2 is None else t * (int(key2.max()) + 2) + (key2 + 1)
    uniq, inv = np.unique(key.reshape(-1), return_inverse=True)
    acc = np.zeros((len(uniq), 3))
    for i in range(3):
        d0 = P[:, (i + 1) % 3] - P[:, i]; d1 = P[:, (i + 2) % 3] - P[:, i]
        d0 /= np.maximum(np.linalg.norm(d0, axis=1, keepdims=True), 1e-300); d1 /= np.maximum(np.linalg.norm(d1, axis=1, keepdims=True), 1e-300)
        ang = np.arccos(np.clip(np.sum(d0 * d1, axis=1), -1.0, 1.0))
        np.add.at(acc, inv.reshape(-1, 3)[:, i], fn * ang[:, None])
    la = np.linalg.norm(acc, axis=1, keepdims=True)
    acc = np.where(la > 0, acc / np.where(la > 0, la, 1.0), 0.0)
    return acc[inv.reshape(-1, 3)]


_PLY_TYPES = {"char": "i1", "uchar": "u1", "short": "i2", "ushort": "u2", "int": "i4", "uint": "u4", "float": "f4",
              "double": "f8", "int8": "i1", "uint8": "u1", "int16": "i2", "uint16": "u2", "int32": "i4", "uint32": "u4",
              "float32": "f4", "float64": "f8"}


def load_ply(path: str, with_attributes: bool = False):
    """Stanford PLY (ascii / binary_little_endian / binary_big_endian) -> (n,3,3) float64 triangle soup:
    vertex x/y/z and the face index list; polygons are fan-triangulated.  ``with_attributes``: also the corner texture
    coordinates (n,6) float32 from the vertex properties ``u v`` / ``s t`` (None without) and the corner normals (n,3,3)
    from ``nx ny nz`` — or, when the file has none, the vertex normals mitsuba computes in that case
    (Mesh::recompute_vertex_normals) [mitsuba3: src/shapes/ply.cpp]; returns (tris, uv, normals).  Other properties are
    skipped."""
    with open(path, "rb") as fh:
        if fh.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, elements = None, []
        while True:
            line = fh.readline()
            if not line:
                raise ValueError(f"{path}: truncated PLY header")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] in ("comment", "obj_info"):
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                elements.append((tok[1], int(tok[2]), []))
            elif tok[0] == "property":
                elements[-1][2].append(tok[1:])
            elif tok[0] == "end_header":
                break
        body = fh.read()
    verts, faces = None, []
    vattr = {}                       # vertex properties beside the position, by name
    if fmt == "ascii":
        lines = body.decode("ascii", "replace").split("\n")
        li = 0
        for name, count, props in elements:
            rows = [lines[li + i].split() for i in range(count)]
            li += count
            if name == "vertex":
                names = [p[-1] for p in props]
                ix = [names.index(a) for a in "xyz"]
                verts = np.asarray([[float(r[i]) for i in ix] for r in rows], dtype=np.float64)
                for a in ("nx", "ny", "nz", "u", "v", "s", "t"):
                    if a in names:
                        vattr[a] = np.asarray([float(r[names.index(a)]) for r in rows], dtype=np.float64)
            elif name == "face":
                for r in rows:
                    n = int(r[0])
                    idx = [int(x) for x in r[1:1 + n]]
                    faces.extend((idx[0], idx[k], idx[k + 1]) for k in range(1, n - 1))
    elif fmt in ("binary_little_endian", "binary_big_endian"):
        e = "<" if fmt == "binary_little_endian" else ">"
        off = 0
        for name, count, props in elements:
            if all(p[0] != "list" for p in props):
                dt = np.dtype([(p[1], e + _PLY_TYPES[p[0]]) for p in props])
                arr = np.frombuffer(body, dtype=dt, count=count, offset=off)
                off += dt.itemsize * count
                if name == "vertex":
                    verts = np.stack([arr["x"], arr["y"], arr["z"]], axis=1).astype(np.float64)
                    for a in ("nx", "ny", "nz", "u", "v", "s", "t"):
                        if a in arr.dtype.names:
                            vattr[a] = arr[a].astype(np.float64)
            else:
                if len(props) != 1:
                    raise ValueError(f"{path}: face element with extra properties is not supported")
                ct, it = np.dtype(e + _PLY_TYPES[props[0][1]]), np.dtype(e + _PLY_TYPES[props[0][2]])
                # fast path: every polygon has the same vertex count
                n0 = int(np.frombuffer(body, dtype=ct, count=1, offset=off)[0]) if count else 0
                rec = np.dtype([("n", ct), ("i", it, (n0,))]) if n0 else None
                ok = False
                if rec is not None and off + rec.itemsize * count <= len(body):
                    arr = np.frombuffer(body, dtype=rec, count=count, offset=off)
                    ok = bool(np.all(arr["n"] == n0))
                if ok:
                    off += rec.itemsize * count
                    if name == "face":
                        idx = arr["i"].astype(np.int64)
                        for k in range(1, n0 - 1):
                            faces.extend(map(tuple, idx[:, [0, k, k + 1]]))
                else:
                    for _ in range(count):
                        n = int(np.frombuffer(body, dtype=ct, count=1, offset=off)[0]); off += ct.itemsize
                        idx = np.frombuffer(body, dtype=it, count=n, offset=off).astype(np.int64); off += it.itemsize * n
                        if name == "face":
                            faces.extend((int(idx[0]), int(idx[k]), int(idx[k + 1])) for k in range(1, n - 1))
    else:
        raise ValueError(f"{path}: unknown PLY format '{fmt}'")
    if verts is None:
        raise ValueError(f"{path}: no vertex element")
    t = np.asarray(faces, dtype=np.int64).reshape(-1, 3)
    if not with_attributes:
        return verts[t]
    uv = None
    for a, b in (("u", "v"), ("s", "t")):
        if a in vattr and b in vattr:
            uv = np.stack([vattr[a], vattr[b]], axis=1).astype(np.float32)[t].reshape(-1, 6)
            break
    if all(a in vattr for a in ("nx", "ny", "nz")):
        normals = np.stack([vattr["nx"], vattr["ny"], vattr["nz"]], axis=1)[t]
    else:
        normals = vertex_normals(verts, t)
    return verts[t], uv, normals


# -- sensor ---------------------------------------------------------------
def perspective_matrices(sd: Dict[str, Any], film_size, crop_size, crop_offset):
    """sample_to_camera and to_world of mitsuba's ``perspective`` sensor
    [mitsuba3: src/sensors/perspective.cpp, include/mitsuba/render/sensor.h parse_fov/perspective_projection]."""
    W, H = float(film_size[0]), float(film_size[1])
    aspect = W / H
    fov = float(sd.get("fov", 0.0))
    if "fov" not in sd:
        if "focal_length" in sd:
            raise ValueError("perspective: 'focal_length' is not supported; pass 'fov'")
        raise ValueError("perspective: 'fov' is required")
    axis = sd.get("fov_axis", "x")
    if axis == "smaller":
        axis = "y" if aspect > 1 else "x"
    elif axis == "larger":
        axis = "x" if aspect > 1 else "y"
    if axis == "y":
        fov = math.degrees(2.0 * math.atan(math.tan(0.5 * math.radians(fov)) * aspect))
    elif axis == "diagonal":
        diag = 2.0 * math.tan(0.5 * math.radians(fov))
        width = diag / math.sqrt(1.0 + 1.0 / (aspect * aspect))
        fov = math.degrees(2.0 * math.atan(width * 0.5))
    elif axis != "x":
        raise ValueError(f"perspective: unknown fov_axis '{axis}'")
    near = float(sd.get("near_clip", 1e-2))
    far = float(sd.get("far_clip", 1e4))
    rel_size = (crop_size[0] / W, crop_size[1] / H)
    rel_off = (crop_offset[0] / W, crop_offset[1] / H)
    T = ScalarTransform4f
    camera_to_sample = (T().scale([1.0 / rel_size[0], 1.0 / rel_size[1], 1.0])
                        .translate([-rel_off[0], -rel_off[1], 0.0])
                        .scale([-0.5, -0.5 * aspect, 1.0])
                        .translate([-1.0, -1.0 / aspect, 0.0])) @ T.perspective(fov, near, far)
    sample_to_camera = camera_to_sample.inverse()
    to_world = to_transform(sd.get("to_world"))
    return sample_to_camera.matrix, to_world.matrix, near, far


class SceneData:
    """Flat, float32 description of a scene: exactly what crosses the C-ABI."""

    def __init__(self):
        self.tri_verts = np.zeros((0, 9), np.float32)
        self.tri_material = np.zeros(0, np.uint32)
        self.tri_emitter = np.zeros(0, np.int32)
        self.materials = (_cabi.mtr_material * 1)()
        self.n_materials = 0
        self.emitters = (_cabi.mtr_emitter * 1)()
        self.n_emitters = 0
        self.camera = _cabi.mtr_camera()
        self.film = _cabi.mtr_film_desc()
        self.shape_names: List[str] = []
        self.shape_ranges: List[tuple] = []
        self.shapes = (_cabi.mtr_shape * 1)()
        self.n_shapes = 0
        self.tri_uv = None               # (n_tris, 6) f32 corner texture coordinates, or None
        self.tri_normals = None          # (n_tris, 9) f32 corner shading normals (all-zero rows: flat triangle), or None
        self.textures = []               # [(H, W, 3) f32 linear RGB] bitmaps referenced by mtr_material.albedo_texture
        self.nlos = None                 # mtr_nlos_desc for the NLOS tier

    def desc(self) -> _cabi.mtr_scene_desc:
        d = _cabi.mtr_scene_desc()
        d.n_tris = self.tri_verts.shape[0]
        d.tri_verts = self.tri_verts.ctypes.data_as(C.POINTER(C.c_float))
        d.tri_material = self.tri_material.ctypes.data_as(C.POINTER(C.c_uint32))
        d.tri_emitter = self.tri_emitter.ctypes.data_as(C.POINTER(C.c_int32))
        d.n_materials = self.n_materials
        d.materials = C.cast(self.materials, C.POINTER(_cabi.mtr_material))
        d.n_emitters = self.n_emitters
        d.emitters = C.cast(self.emitters, C.POINTER(_cabi.mtr_emitter))
        d.camera = self.camera
        d.film = self.film
        d.n_shapes = self.n_shapes
        d.shapes = C.cast(self.shapes, C.POINTER(_cabi.mtr_shape)) if self.n_shapes else None
        d.tri_uv = self.tri_uv.ctypes.data_as(C.POINTER(C.c_float)) if self.tri_uv is not None else None
        d.tri_normals = self.tri_normals.ctypes.data_as(C.POINTER(C.c_float)) if self.tri_normals is not None else None
        if self.textures:
            self._tex_desc = (_cabi.mtr_texture * len(self.textures))()
            for i, t in enumerate(self.textures):
                self._tex_desc[i].height, self._tex_desc[i].width = int(t.shape[0]), int(t.shape[1])
                self._tex_desc[i].rgb = t.ctypes.data_as(C.POINTER(C.c_float))
            d.n_textures, d.textures = len(self.textures), C.cast(self._tex_desc, C.POINTER(_cabi.mtr_texture))
        if self.nlos is not None:
            self.nlos.n_shapes = self.n_shapes
            self.nlos.shapes = C.cast(self.shapes, C.POINTER(_cabi.mtr_shape))
            d.nlos = C.pointer(self.nlos)
        d._keepalive = self            # arrays must outlive the desc
        return d


def film_desc_from(film) -> _cabi.mtr_film_desc:
    f = _cabi.mtr_film_desc()
    f.width, f.height = int(film.size_[0]), int(film.size_[1])
    f.crop_width, f.crop_height = int(film.crop_size_[0]), int(film.crop_size_[1])
    f.crop_offset_x, f.crop_offset_y = int(film.crop_offset_[0]), int(film.crop_offset_[1])
    f.temporal_bins = int(film.temporal_bins)
    f.start_opl = np.float32(film.start_opl)
    f.bin_width_opl = np.float32(film.bin_width_opl)
    if getattr(film, "exhaustive_scan", False):
        f.laser_scan_width, f.laser_scan_height = int(film.laser_scan_width), int(film.laser_scan_height)
    fr = getattr(film, "frequencies_f32", None)                 # phasor_hdr_film
    if fr is not None:
        f.n_frequencies = int(fr.size)
        f.frequencies = fr.ctypes.data_as(C.POINTER(C.c_float))
        f._keepalive = fr
    return f


def nlos_desc_from(integrator, sensor, emitter, relay_shape: int) -> _cabi.mtr_nlos_desc:
    """mtr_nlos_desc from the live plugin objects (so that nlos.focus_emitter_* edits are picked up)."""
    n = _cabi.mtr_nlos_desc()
    origin = getattr(sensor, "sensor_origin", (0.0, 0.0, 0.0))       # perspective sensor: unused
    for k in range(3):
        n.sensor_origin[k] = np.float32(origin[k])
        n.laser_irradiance[k] = np.float32(emitter.irradiance[k])
    n.relay_shape = relay_shape if relay_shape >= 0 else _cabi.MTR_NLOS_NO_RELAY
    m = emitter.world_transform().matrix.reshape(-1)
    for i in range(16):
        n.laser_to_world[i] = np.float32(m[i])
    n.laser_fov = np.float32(emitter.fov)
    n.laser_scale = np.float32(emitter.scale)
    n.capture_type = int(integrator.capture_type)
    n.flags = int(integrator.nlos_flags())
    n.filter_depth = int(integrator.filter_depth)
    n.illumination_scan_fov = np.float32(integrator.illumination_scan_fov)
    n.sensor_is_confocal = 1 if getattr(sensor, "is_confocal", False) else 0
    tgt = np.asarray(getattr(sensor, "laser_target", (0.0, 0.0, 0.0)), dtype=np.float64).reshape(3)
    for k in range(3):
        n.sensor_target[k] = np.float32(tgt[k])
    return n


def save_geometry(sd: "SceneData", path: str, **meta):
    """flattened geometry + material / emitter tables of a SceneData -> one .npz (a data fixture that can travel
    where the scene's asset files cannot); sensor / film / integrator are NOT stored — the caller supplies them"""
    import json
    np.savez_compressed(
        path, tri_verts=sd.tri_verts, tri_material=sd.tri_material.astype(np.uint16 if sd.n_materials < 65536 else np.uint32),
        tri_emitter=sd.tri_emitter.astype(np.int16),
        materials=np.frombuffer(bytes(sd.materials), dtype=np.uint8)[:sd.n_materials * C.sizeof(_cabi.mtr_material)],
        emitters=np.frombuffer(bytes(sd.emitters), dtype=np.uint8)[:sd.n_emitters * C.sizeof(_cabi.mtr_emitter)],
        shapes=np.frombuffer(bytes(sd.shapes), dtype=np.uint8)[:sd.n_shapes * C.sizeof(_cabi.mtr_shape)],
        tri_uv=sd.tri_uv if sd.tri_uv is not None else np.zeros((0, 6), np.float32),
        tri_normals=sd.tri_normals if sd.tri_normals is not None else np.zeros((0, 9), np.float32),
        layout=np.asarray([C.sizeof(_cabi.mtr_material), C.sizeof(_cabi.mtr_emitter), C.sizeof(_cabi.mtr_shape)]),
        meta=np.asarray(json.dumps(meta)))


def load_geometry(path: str) -> Dict[str, Any]:
    import json
    z = np.load(path)
    msize = C.sizeof(_cabi.mtr_material)
    mat_bytes = z["materials"]
    layout = list(z["layout"])
    if layout[0] == 64 and msize > 64:           # written with ABI <= 7: mtr_material grew by appended fields (rough lobes) only
        mat_bytes = np.pad(mat_bytes.reshape(-1, 64), ((0, 0), (0, msize - 64))).reshape(-1)
        layout[0] = msize
    em_bytes = z["emitters"]
    esize = C.sizeof(_cabi.mtr_emitter)
    if layout[1] == 60 and esize > 60:           # written with ABI <= 8: mtr_emitter grew by an appended field (flip_normals) only
        em_bytes = np.pad(em_bytes.reshape(-1, 60), ((0, 0), (0, esize - 60))).reshape(-1)
        layout[1] = esize
    if layout != [msize, C.sizeof(_cabi.mtr_emitter), C.sizeof(_cabi.mtr_shape)]:
        raise ValueError(f"{path}: material / emitter record sizes {list(z['layout'])} do not match this C-ABI; "
                         "regenerate with tests/golden/make_golden.py")
    z = dict(z.items()); z["materials"] = mat_bytes; z["emitters"] = em_bytes
    nm = z["materials"].size // C.sizeof(_cabi.mtr_material)
    ne = z["emitters"].size // C.sizeof(_cabi.mtr_emitter)
    mats = (_cabi.mtr_material * max(1, nm)).from_buffer_copy(z["materials"].tobytes().ljust(C.sizeof(_cabi.mtr_material), b"\0"))
    ems = (_cabi.mtr_emitter * max(1, ne)).from_buffer_copy(z["emitters"].tobytes().ljust(C.sizeof(_cabi.mtr_emitter), b"\0"))
    ns = z["shapes"].size // C.sizeof(_cabi.mtr_shape)
    shapes = (_cabi.mtr_shape * max(1, ns)).from_buffer_copy(z["shapes"].tobytes().ljust(C.sizeof(_cabi.mtr_shape), b"\0"))
    return {"shapes": shapes, "n_shapes": ns,
            "tri_uv": np.ascontiguousarray(z["tri_uv"], dtype=np.float32) if z["tri_uv"].shape[0] else None,
            "tri_normals": (np.ascontiguousarray(z["tri_normals"], dtype=np.float32)
                            if "tri_normals" in z and z["tri_normals"].shape[0] else None),
            "tri_verts": np.ascontiguousarray(z["tri_verts"], dtype=np.float32),
            "tri_material": np.ascontiguousarray(z["tri_material"].astype(np.uint32)),
            "tri_emitter": np.ascontiguousarray(z["tri_emitter"].astype(np.int32)),
            "materials": mats, "n_materials": nm, "emitters": ems, "n_emitters": ne,
            "meta": json.loads(str(z["meta"]))}


def flatten_scene(d: Dict[str, Any], film, sensor_dict: Dict[str, Any], base_dir: str = ".",
                  relay_shape_name: Optional[str] = None, approximate_materials: bool = False,
                  geometry: Optional[Dict[str, Any]] = None) -> SceneData:
    b = _SceneBuilder(d, base_dir, approximate_materials)
    for name, v in d.items():
        if not isinstance(v, dict):
            continue
        t = v.get("type")
        if t in ("rectangle", "cube", "obj", "ply", "sphere", "disk", "cylinder"):
            b.add_shape(name, v)
    sd = SceneData()
    if geometry is not None:                     # pre-flattened geometry (load_geometry); the dictionary has no shapes
        if b.tri_verts:
            raise ValueError("flatten_scene: pre-flattened geometry cannot be mixed with shape plugins")
        sd.tri_verts, sd.tri_material, sd.tri_emitter = geometry["tri_verts"], geometry["tri_material"], geometry["tri_emitter"]
        b.materials, b.emitters = list(geometry["materials"])[:geometry["n_materials"]], list(geometry["emitters"])[:geometry["n_emitters"]]
        b.shapes = list(geometry["shapes"])[:geometry["n_shapes"]]
        sd.tri_uv = geometry["tri_uv"]
        sd.tri_normals = geometry.get("tri_normals")
        sd.textures = [np.ascontiguousarray(t, dtype=np.float32) for t in geometry.get("textures", [])]
        if _variant.is_monochromatic():
            # the records were flattened from RGB values: the monochromatic variants replace every colour by its luminance at
            # load time (_color3), so the same happens to the stored tables (copies: the fixture stays as it is)
            def lum3(v):
                c = [np.float32(v[0]), np.float32(v[1]), np.float32(v[2])]
                return (c[0] * np.float32(0.212671) + c[1] * np.float32(0.715160)) + c[2] * np.float32(0.072169)
            mats, ems = [], []
            for m in b.materials:
                m = type(m).from_buffer_copy(m)
                aniso = bool(m.flags & _cabi.MTR_MAT_ANISOTROPIC)            # alpha_v travels in c2[0] (roughconductor) / b[0] (roughdielectric)
                keep = ("b" if m.type == _cabi.MTR_BSDF_ROUGHDIELECTRIC else "c2") if aniso else None
                alpha_v = getattr(m, keep)[0] if keep else None
                for fld in ("a", "b", "c", "c2"):
                    arr = getattr(m, fld)
                    arr[0] = arr[1] = arr[2] = lum3(arr)
                if keep:
                    getattr(m, keep)[0] = alpha_v
                if m.type in (_cabi.MTR_BSDF_ROUGHPLASTIC, _cabi.MTR_BSDF_PLASTIC) and (m.a[0] + m.c[0]) > 0:
                    m.specular_sampling_weight = np.float32(m.c[0] / (m.a[0] + m.c[0]))
                mats.append(m)
            for e in b.emitters:
                e = type(e).from_buffer_copy(e)
                e.radiance[0] = e.radiance[1] = e.radiance[2] = lum3(e.radiance)
                ems.append(e)
            b.materials, b.emitters = mats, ems
            sd.textures = [np.repeat(((t[..., 0] * np.float32(0.212671) + t[..., 1] * np.float32(0.715160)) + t[..., 2] * np.float32(0.072169))[..., None], 3, axis=-1)
                           for t in sd.textures]
    if b.tri_verts:
        sd.tri_verts = np.ascontiguousarray(np.concatenate(b.tri_verts).reshape(-1, 9))
        sd.tri_material = np.ascontiguousarray(np.concatenate(b.tri_mat))
        sd.tri_emitter = np.ascontiguousarray(np.concatenate(b.tri_em))
        sd.textures = [np.ascontiguousarray(t, dtype=np.float32) for t in b.textures]
        if any(u is not None for u in b.tri_normals):   # flat shapes: all-zero normals
            sd.tri_normals = np.ascontiguousarray(np.concatenate(
                [u if u is not None else np.zeros((v.shape[0], 9), np.float32) for u, v in zip(b.tri_normals, b.tri_verts)]))
        if any(u is not None for u in b.tri_uv):        # shapes without texture coordinates: a degenerate (all-zero) parameterisation
            sd.tri_uv = np.ascontiguousarray(np.concatenate(
                [u if u is not None else np.zeros((v.shape[0], 6), np.float32) for u, v in zip(b.tri_uv, b.tri_verts)]))
    sd.n_materials = len(b.materials)
    sd.materials = (_cabi.mtr_material * max(1, sd.n_materials))(*b.materials)
    sd.n_emitters = len(b.emitters)
    sd.emitters = (_cabi.mtr_emitter * max(1, sd.n_emitters))(*b.emitters)
    sd.n_shapes = len(b.shapes)
    sd.shapes = (_cabi.mtr_shape * max(1, sd.n_shapes))(*b.shapes)
    if sensor_dict.get("type") == "perspective":
        s2c, tw, near, far = perspective_matrices(sensor_dict, film.size_, film.crop_size_, film.crop_offset_)
        for i in range(16):
            sd.camera.sample_to_camera[i] = np.float32(s2c.reshape(-1)[i])
            sd.camera.to_world[i] = np.float32(tw.reshape(-1)[i])
        sd.camera.near_clip, sd.camera.far_clip = np.float32(near), np.float32(far)
    sd.film = film_desc_from(film)
    sd.shape_names, sd.shape_ranges = b.shape_names, b.shape_ranges
    sd.relay_shape = b.shape_names.index(relay_shape_name) if relay_shape_name is not None else -1
    return sd
