"""ctypes mirror of ``include/mitransient_amd.h`` and the loader of the HIP library.

There is NO CPU fallback: if ``libmitransient_amd.so`` (the gfx950 HIP build) is
missing or no HIP device is visible, every product entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MITRANSIENT_AMD_LIB") or os.path.join(_HERE, "csrc", "libmitransient_amd.so")   # env override: kernel A/B experiments

MTR_ABI_VERSION = 12
MTR_TRAIT_DIFFUSE, MTR_TRAIT_ONE_RECT_EMITTER, MTR_TRAIT_LEAF_PAIR, MTR_TRAIT_FLAT_TOP, MTR_TRAIT_FLAT_LEAVES, MTR_TRAIT_NO_LOBES, MTR_TRAIT_GREY = 1, 2, 4, 8, 16, 32, 64      # mtr_scene_traits
MTR_SPLAT_FILM_ZERO = 0x100      # mtr_splat_add: OR into `variant` when the film is all-zero on entry

MTR_BSDF_DIFFUSE, MTR_BSDF_CONDUCTOR, MTR_BSDF_DIELECTRIC, MTR_BSDF_NULL = 0, 1, 2, 3
MTR_BSDF_ROUGHCONDUCTOR, MTR_BSDF_ROUGHPLASTIC, MTR_BSDF_ROUGHDIELECTRIC, MTR_BSDF_THINDIELECTRIC, MTR_BSDF_PLASTIC = 4, 5, 6, 7, 8
MTR_MAT_TWOSIDED, MTR_MAT_NONLINEAR, MTR_MAT_BECKMANN, MTR_MAT_ANISOTROPIC = 1, 2, 4, 8
MTR_ROUGH_TRANSMITTANCE_RES = 64
MTR_FLAG_CAMERA_UNWARP = 1
MTR_FLAG_DISCARD_DIRECT_LIGHT = 2
MTR_FLAG_FILM_ZERO = 4
MTR_FLAG_PCG_INITSEQ_PLUS_LANE = 8
MTR_FLAG_PCG_TEA64 = 128
MTR_FLAG_KEEP_COUNTERS = 16
MTR_FLAG_DETERMINISTIC = 32
MTR_FLAG_DEVELOPED_ROWS = 64
MTR_MODE_AUTO, MTR_MODE_FUSED, MTR_MODE_WAVEFRONT = 0, 1, 2
MTR_RECT_ANALYTIC, MTR_RECT_FLIP_NORMALS = 1, 2

_f3 = C.c_float * 3
_f16 = C.c_float * 16


class mtr_material(C.Structure):
    _fields_ = [("type", C.c_uint32), ("flags", C.c_uint32),
                ("a", _f3), ("b", _f3), ("c", _f3),
                ("int_ior", C.c_float), ("ext_ior", C.c_float), ("c2", _f3),
                ("alpha", C.c_float), ("internal_reflectance", C.c_float), ("specular_sampling_weight", C.c_float),
                ("albedo_texture", C.c_uint32), ("external_transmittance", C.c_float * MTR_ROUGH_TRANSMITTANCE_RES)]


class mtr_texture(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("rgb", C.POINTER(C.c_float))]


class mtr_emitter(C.Structure):
    _fields_ = [("center", _f3), ("du", _f3), ("dv", _f3), ("radiance", _f3),
                ("is_mesh", C.c_uint32), ("first_tri", C.c_uint32), ("n_tris", C.c_uint32),
                ("flip_normals", C.c_uint32)]


class mtr_camera(C.Structure):
    _fields_ = [("sample_to_camera", _f16), ("to_world", _f16),
                ("near_clip", C.c_float), ("far_clip", C.c_float)]


class mtr_film_desc(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32),
                ("crop_width", C.c_uint32), ("crop_height", C.c_uint32),
                ("crop_offset_x", C.c_uint32), ("crop_offset_y", C.c_uint32),
                ("temporal_bins", C.c_uint32),
                ("start_opl", C.c_float), ("bin_width_opl", C.c_float),
                ("laser_scan_width", C.c_uint32), ("laser_scan_height", C.c_uint32),
                ("n_frequencies", C.c_uint32), ("frequencies", C.POINTER(C.c_float))]


MTR_CAPTURE_SINGLE, MTR_CAPTURE_CONFOCAL, MTR_CAPTURE_EXHAUSTIVE = 1, 2, 3
MTR_NLOS_FORCE_EQUAL_GRIDS = 64
MTR_NLOS_NO_RELAY = 0xFFFFFFFF
MTR_NLOS_LASER_SAMPLING, MTR_NLOS_HG_SAMPLING, MTR_NLOS_HG_RROULETTE = 1, 2, 4
MTR_NLOS_HG_INCLUDES_WALL, MTR_NLOS_ACCOUNT_FIRST_LAST, MTR_NLOS_DISCARD_DIRECT = 8, 16, 32


class mtr_shape(C.Structure):
    _fields_ = [("first_tri", C.c_uint32), ("n_tris", C.c_uint32), ("is_rectangle", C.c_uint32),
                ("center", _f3), ("du", _f3), ("dv", _f3), ("has_to_world", C.c_uint32), ("to_world", C.c_float * 12)]


class mtr_nlos_desc(C.Structure):
    _fields_ = [("sensor_origin", _f3), ("relay_shape", C.c_uint32), ("laser_to_world", _f16),
                ("laser_fov", C.c_float), ("laser_irradiance", _f3), ("laser_scale", C.c_float),
                ("capture_type", C.c_uint32), ("flags", C.c_uint32), ("filter_depth", C.c_int32),
                ("illumination_scan_fov", C.c_float), ("n_shapes", C.c_uint32), ("shapes", C.POINTER(mtr_shape)),
                ("sensor_is_confocal", C.c_uint32), ("sensor_target", _f3)]


class mtr_scene_desc(C.Structure):
    _fields_ = [("n_tris", C.c_uint32),
                ("tri_verts", C.POINTER(C.c_float)),
                ("tri_material", C.POINTER(C.c_uint32)),
                ("tri_emitter", C.POINTER(C.c_int32)),
                ("n_materials", C.c_uint32),
                ("materials", C.POINTER(mtr_material)),
                ("n_emitters", C.c_uint32),
                ("emitters", C.POINTER(mtr_emitter)),
                ("camera", mtr_camera),
                ("film", mtr_film_desc),
                ("nlos", C.POINTER(mtr_nlos_desc)),
                ("n_shapes", C.c_uint32),
                ("shapes", C.POINTER(mtr_shape)),
                ("tri_uv", C.POINTER(C.c_float)),
                ("tri_normals", C.POINTER(C.c_float)),
                ("n_textures", C.c_uint32),
                ("textures", C.POINTER(mtr_texture))]


class mtr_render_params(C.Structure):
    _fields_ = [("spp_total", C.c_uint32), ("spp_begin", C.c_uint32), ("spp_end", C.c_uint32),
                ("pixel_begin", C.c_uint32), ("pixel_end", C.c_uint32),
                ("seed", C.c_uint32), ("max_depth", C.c_int32), ("rr_depth", C.c_int32),
                ("flags", C.c_uint32), ("mode", C.c_uint32), ("spp_scale", C.c_uint32), ("reserve_cus", C.c_uint32),
                ("n_bands", C.c_uint32), ("band_epoch", C.c_uint32), ("band_done", C.c_uint64)]


class mtr_counters(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("rays_closest", C.c_uint64), ("rays_shadow", C.c_uint64),
                ("splats_issued", C.c_uint64), ("bounces", C.c_uint64), ("splats_overflow", C.c_uint64),
                ("reserved", C.c_uint64 * 2)]

    def as_dict(self):
        d = {k: int(getattr(self, k)) for k in
             ("paths", "rays_closest", "rays_shadow", "splats_issued", "bounces", "splats_overflow")}
        d["reserved"] = [int(self.reserved[0]), int(self.reserved[1])]
        return d


class mtr_splat_soa(C.Structure):
    _fields_ = [("pixel", C.c_void_p), ("opl", C.c_void_p),
                ("r", C.c_void_p), ("g", C.c_void_p), ("b", C.c_void_p), ("n", C.c_uint64),
                ("laser", C.c_void_p)]


class mtr_kernel_times(C.Structure):
    _fields_ = [("total_ms", C.c_float), ("trace_ms", C.c_float), ("scatter_ms", C.c_float),
                ("trace_launches", C.c_uint32), ("scatter_launches", C.c_uint32),
                ("wf_trace_ms", C.c_float), ("wf_trace_kernel_launches", C.c_uint32),
                ("wf_shade_ms", C.c_float)]

    def as_dict(self):
        return {"total_ms": float(self.total_ms), "trace_ms": float(self.trace_ms),
                "scatter_ms": float(self.scatter_ms), "trace_launches": int(self.trace_launches),
                "scatter_launches": int(self.scatter_launches), "wf_trace_ms": float(self.wf_trace_ms),
                "wf_trace_kernel_launches": int(self.wf_trace_kernel_launches), "wf_shade_ms": float(self.wf_shade_ms)}


# Every symbol include/mitransient_amd.h declares (checked by tests/test_abi.py).
EXPORTS = [
    "mtr_abi_version", "mtr_ctx_create", "mtr_ctx_destroy", "mtr_ctx_set_stream", "mtr_last_error",
    "mtr_scene_create", "mtr_scene_destroy", "mtr_scene_set_film", "mtr_scene_set_nlos", "mtr_scene_bvh_info", "mtr_scene_traits",
    "mtr_ctx_trim", "mtr_film_clear", "mtr_render", "mtr_render_plan", "mtr_counters_reset", "mtr_counters_read", "mtr_film_develop", "mtr_splat_add", "mtr_debug_set_splat_log",
]

_lib = None


class MitransientAMDError(RuntimeError):
    pass


def load_library() -> C.CDLL:
    """Load the HIP library (built by ``__graft_entry__.build()``). Fails loudly."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MitransientAMDError(
            f"{LIB_PATH} not found: build the gfx950 HIP library first "
            "(python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback.")
    # torch first: it ships its own libamdhip64; loading ours afterwards makes the dynamic linker
    # resolve the same soname to the runtime torch already initialised, so device pointers and
    # streams are shared (two HIP runtimes in one process do not see each other's devices).
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    lib.mtr_abi_version.restype = C.c_int
    lib.mtr_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    lib.mtr_ctx_destroy.argtypes = [vp]
    lib.mtr_ctx_destroy.restype = None
    lib.mtr_ctx_set_stream.argtypes = [vp, vp]
    lib.mtr_last_error.argtypes = [vp]
    lib.mtr_last_error.restype = C.c_char_p
    lib.mtr_scene_create.argtypes = [vp, C.POINTER(mtr_scene_desc), C.POINTER(vp)]
    lib.mtr_scene_destroy.argtypes = [vp]
    lib.mtr_scene_destroy.restype = None
    lib.mtr_scene_set_film.argtypes = [vp, C.POINTER(mtr_film_desc)]
    lib.mtr_scene_set_nlos.argtypes = [vp, C.POINTER(mtr_nlos_desc)]
    lib.mtr_scene_bvh_info.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.mtr_scene_traits.argtypes = [vp, C.POINTER(C.c_uint32)]
    lib.mtr_counters_read.argtypes = [vp, C.POINTER(mtr_counters)]
    lib.mtr_counters_reset.argtypes = [vp]
    lib.mtr_ctx_trim.argtypes = [vp]
    lib.mtr_render_plan.argtypes = [vp, C.POINTER(mtr_render_params), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.mtr_film_clear.argtypes = [vp, C.POINTER(mtr_film_desc), vp, vp]
    lib.mtr_render.argtypes = [vp, C.POINTER(mtr_render_params), vp, vp,
                               C.POINTER(mtr_counters), C.POINTER(mtr_kernel_times)]
    lib.mtr_film_develop.argtypes = [vp, C.POINTER(mtr_film_desc), vp, vp, vp, vp]
    lib.mtr_splat_add.argtypes = [vp, C.POINTER(mtr_splat_soa), C.POINTER(mtr_film_desc), C.c_int, vp,
                                  C.POINTER(C.c_float)]
    lib.mtr_debug_set_splat_log.argtypes = [vp, vp, C.c_uint64, vp]
    if lib.mtr_abi_version() != MTR_ABI_VERSION:
        raise MitransientAMDError("libmitransient_amd.so ABI version mismatch; rebuild")
    _lib = lib
    return lib


def check(status: int, ctx=None, what: str = ""):
    if status == 0:
        return
    msg = ""
    try:
        msg = (load_library().mtr_last_error(ctx) or b"").decode()
    except Exception:  # pragma: no cover
        pass
    raise MitransientAMDError(f"{what} failed (status {status}): {msg}")
