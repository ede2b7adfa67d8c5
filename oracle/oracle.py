"""ctypes wrapper of the CPU oracle (TEST INFRASTRUCTURE — never imported by mitransient_amd/).

Only tests/, ``__graft_entry__.smoke()`` and bench.py's ``cpu_baseline`` leg load this.
PARITY UNPINNED against real Mitsuba (see mtr_oracle.c header).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from mitransient_amd import _cabi   # POD struct definitions of the boundary only

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "libmtr_oracle.so")
_lib = None


class orc_splat_rec(C.Structure):
    _fields_ = [("lane", C.c_uint32), ("depth_kind", C.c_uint32), ("pixel", C.c_uint32), ("bin", C.c_uint32),
                ("r", C.c_float), ("g", C.c_float), ("b", C.c_float), ("opl", C.c_float)]


def build(force=False):
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(os.path.join(_HERE, "mtr_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return LIB


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        L = C.CDLL(LIB)
        fp, u32p = C.POINTER(C.c_float), C.POINTER(C.c_uint32)
        L.orc_render.argtypes = [C.POINTER(_cabi.mtr_scene_desc), C.POINTER(_cabi.mtr_render_params), fp, fp,
                                 C.POINTER(_cabi.mtr_counters), C.c_int, C.c_int,
                                 C.POINTER(orc_splat_rec), C.c_uint64, C.POINTER(C.c_uint64)]
        L.orc_render.restype = C.c_int
        L.orc_develop.argtypes = [C.POINTER(_cabi.mtr_film_desc), fp, fp, fp, fp]
        L.orc_develop.restype = None
        L.orc_splat_add.argtypes = [C.POINTER(_cabi.mtr_film_desc), C.c_uint64, u32p, fp, fp, fp, fp, fp, u32p, u32p]
        L.orc_splat_add.restype = None
        L.orc_bin_index.argtypes = [C.c_float, C.c_float, C.c_float, C.c_uint32]
        L.orc_bin_index.restype = C.c_int
        L.orc_intersect.argtypes = [C.POINTER(_cabi.mtr_scene_desc), C.c_uint32, fp, fp, fp, C.c_int, fp,
                                    C.POINTER(C.c_int32), C.POINTER(C.c_uint8)]
        L.orc_intersect.restype = None
        L.orc_camera_ray.argtypes = [C.POINTER(_cabi.mtr_scene_desc), C.c_uint32, C.c_uint32, C.c_float, C.c_float, fp, fp, fp]
        L.orc_camera_ray.restype = None
        L.orc_pcg32_stream.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, u32p, fp]
        L.orc_pcg32_stream.restype = None
        L.orc_sampler_stream.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, fp]
        L.orc_sampler_stream.restype = None
        L.orc_tea32.argtypes = [C.c_uint32, C.c_uint32, C.c_int, u32p]
        L.orc_tea32.restype = None
        L.orc_sincos_q.argtypes = [C.c_float, fp, fp]
        L.orc_sincos_q.restype = None
        L.orc_square_to_cos_hemi.argtypes = [C.c_float, C.c_float, fp]
        L.orc_square_to_cos_hemi.restype = None
        L.orc_num_threads.restype = C.c_int
        L.orc_phasor_term.argtypes = [C.c_float, C.c_float, fp, fp]
        L.orc_phasor_term.restype = None
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def film_shape(f):
    """raw tensor shape: (H,W,T,4), or (H,W,Lh,Lw,T,4) for an exhaustive_scan film (transient_image_block.py:63-68)"""
    if f.n_frequencies:                         # phasor_hdr_film: (H, W, 2F+1)
        return (f.height, f.width, 2 * f.n_frequencies + 1)
    if f.laser_scan_width and f.laser_scan_height:
        return (f.height, f.width, f.laser_scan_height, f.laser_scan_width, f.temporal_bins, 4)
    return (f.height, f.width, f.temporal_bins, 4)


def alloc_film(film_desc, prefault=False):
    f = film_desc
    t4 = np.zeros(film_shape(f), np.float32)
    s4 = np.zeros((f.height, f.width, 4), np.float32)
    if prefault:          # touch every page now (np.zeros maps lazily): keeps page faults out of timed regions
        t4.fill(0.0)
        s4.fill(0.0)
    return t4, s4


def render(scene_data, params: _cabi.mtr_render_params, n_threads=0, use_bvh=False, log_capacity=0, out=None):
    """Returns (transient (H,W,T,4) f32, steady (H,W,4) f32, counters dict[, log ndarray]).
    ``out=(t4, s4)`` accumulates into existing buffers."""
    f = scene_data.film
    t4, s4 = out if out is not None else alloc_film(f)
    cnt = _cabi.mtr_counters()
    d = scene_data.desc()
    log = None
    n_log = C.c_uint64(0)
    if log_capacity:
        log = (orc_splat_rec * log_capacity)()
    rc = lib().orc_render(C.byref(d), C.byref(params), _fp(t4), _fp(s4), C.byref(cnt), n_threads, int(use_bvh),
                          log, log_capacity, C.byref(n_log))
    if rc != 0:
        raise RuntimeError(f"orc_render failed ({rc})")
    if log_capacity:
        n = min(int(n_log.value), log_capacity)
        arr = np.frombuffer(log, dtype=np.dtype([("lane", "u4"), ("depth_kind", "u4"), ("pixel", "u4"), ("bin", "u4"),
                                                  ("r", "f4"), ("g", "f4"), ("b", "f4"), ("opl", "f4")]))[:n].copy()
        return t4, s4, cnt.as_dict(), arr
    return t4, s4, cnt.as_dict()


def develop(film_desc, t4=None, s4=None):
    t3 = s3 = None
    if t4 is not None and film_desc.n_frequencies:
        t3 = np.empty(t4.shape[:-1] + (film_desc.n_frequencies, 2), np.float32)
    elif t4 is not None:
        t3 = np.empty(t4.shape[:-1] + (3,), np.float32)
    if s4 is not None:
        s3 = np.empty(s4.shape[:-1] + (3,), np.float32)
    lib().orc_develop(C.byref(film_desc), _fp(t4) if t4 is not None else None, _fp(t3) if t3 is not None else None,
                      _fp(s4) if s4 is not None else None, _fp(s3) if s3 is not None else None)
    # the reference's steady hdrfilm carries the crop window (transient_hdr_film.py:131-144): steady.develop() is
    # (crop_h, crop_w, 3); the accumulator is full-size with the window at its top-left corner
    if s3 is not None and (film_desc.crop_width, film_desc.crop_height) != (film_desc.width, film_desc.height) \
            and s3.shape[:2] == (film_desc.height, film_desc.width):
        s3 = np.ascontiguousarray(s3[:film_desc.crop_height, :film_desc.crop_width])
    return t3, s3


def splat_add(film_desc, pixel, opl, r, g, b, t4, laser_x=None, laser_y=None):
    pixel = np.ascontiguousarray(pixel, np.uint32)
    arrs = [np.ascontiguousarray(x, np.float32) for x in (opl, r, g, b)]
    up = C.POINTER(C.c_uint32)
    lx = np.ascontiguousarray(laser_x, np.uint32) if laser_x is not None else None
    ly = np.ascontiguousarray(laser_y, np.uint32) if laser_y is not None else None
    lib().orc_splat_add(C.byref(film_desc), len(pixel), pixel.ctypes.data_as(up), *[_fp(a) for a in arrs], _fp(t4),
                        lx.ctypes.data_as(up) if lx is not None else None, ly.ctypes.data_as(up) if ly is not None else None)


def phasor_term(freq, opl):
    c, s = C.c_float(0), C.c_float(0)
    lib().orc_phasor_term(np.float32(freq), np.float32(opl), C.byref(c), C.byref(s))
    return c.value, s.value


def bin_index(distance, start, width, T):
    return lib().orc_bin_index(np.float32(distance), np.float32(start), np.float32(width), T)


def intersect(scene_data, o, d, maxt=None, use_bvh=False):
    o = np.ascontiguousarray(o, np.float32)
    d = np.ascontiguousarray(d, np.float32)
    n = o.shape[0]
    t = np.empty(n, np.float32)
    prim = np.empty(n, np.int32)
    occ = np.empty(n, np.uint8)
    mt = np.ascontiguousarray(maxt, np.float32) if maxt is not None else None
    desc = scene_data.desc()
    lib().orc_intersect(C.byref(desc), n, _fp(o), _fp(d), _fp(mt) if mt is not None else None, int(use_bvh),
                        _fp(t), prim.ctypes.data_as(C.POINTER(C.c_int32)), occ.ctypes.data_as(C.POINTER(C.c_uint8)))
    return t, prim, occ


def camera_ray(scene_data, px, py, j1, j2):
    o = np.empty(3, np.float32)
    d = np.empty(3, np.float32)
    mt = C.c_float()
    desc = scene_data.desc()
    lib().orc_camera_ray(C.byref(desc), px, py, j1, j2, _fp(o), _fp(d), C.byref(mt))
    return o, d, mt.value


def pcg32_stream(initstate, initseq, n):
    u = np.empty(n, np.uint32)
    f = np.empty(n, np.float32)
    lib().orc_pcg32_stream(initstate, initseq, n, u.ctypes.data_as(C.POINTER(C.c_uint32)), _fp(f))
    return u, f


def sampler_stream(seed_value, lane, n):
    f = np.empty(n, np.float32)
    lib().orc_sampler_stream(seed_value, lane, n, _fp(f))
    return f


def tea32(v0, v1, rounds=4):
    out = np.empty(2, np.uint32)
    lib().orc_tea32(v0, v1, rounds, out.ctypes.data_as(C.POINTER(C.c_uint32)))
    return int(out[0]), int(out[1])


def num_threads():
    return lib().orc_num_threads()
