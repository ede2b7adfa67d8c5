"""``TransientADIntegrator``: host orchestration of a transient render.

Mirrors mitransient/integrators/common.py — ``__init__`` (:22-30), ``prepare`` (:32-85),
``render`` (:122-213), ``add_transient_f`` (:411-422), ``check_transient_`` (:424-447) —
with the Dr.Jit trace replaced by launches of the HIP library: ``sample_rays`` +
``sample`` + the film splats of one pass are ONE call to ``mtr_render``.
The AD entry points (render_forward / render_backward, :215-409) are out of scope.
"""
from __future__ import annotations

import ctypes as C
from typing import Tuple

from .. import _cabi
from ..films.transient_hdr_film import TransientHDRFilm
from ..runtime import get_context
from ..scene import Properties
from ..tensor import TensorXf


class TransientADIntegrator:
    def __init__(self, props: Properties):
        # [mitsuba3: ADIntegrator.__init__] max_depth default 6 (-1 = infinite), rr_depth default 5
        max_depth = int(props.get("max_depth", 6))
        if max_depth < 0 and max_depth != -1:
            raise Exception("\"max_depth\" must be set to -1 (infinite) or a value >= 0")
        self.max_depth = max_depth if max_depth != -1 else 0xFFFFFFFF
        self.rr_depth = int(props.get("rr_depth", 5))
        if self.rr_depth <= 0:
            raise Exception("\"rr_depth\" must be set to a value greater than zero!")
        # hide_emitters: transientpath.py reads it in its "Hide the environment emitter" block after si.bsdf(ray) — a camera ray
        # (depth 0) that MISSES the scene stops being active when the flag is set, which only changes what an ENVIRONMENT emitter
        # would have contributed.  The subset has no environment emitter (a miss contributes nothing either way): parsed, kept,
        # without effect here
        self.hide_emitters = bool(props.get("hide_emitters", False))
        # common.py:25-30
        self.camera_unwarp = bool(props.get("camera_unwarp", False))
        self.discard_direct_light = bool(props.get("discard_direct_light", False))
        _ = props.get("gaussian_stddev", 0.5)      # accepted and ignored, as in the reference
        _ = props.get("temporal_filter", "")
        _ = props.get("block_size", 0)
        # sampler seeding variant (extension; see MTR_FLAG_PCG_INITSEQ_PLUS_LANE in the header): False = TEA(seed, lane) only
        self.pcg_initseq_plus_lane = bool(props.get("amd_pcg_initseq_plus_lane", False))
        # ... and the 64-bit reading of the same call site (MTR_FLAG_PCG_TEA64)
        self.pcg_tea64 = bool(props.get("amd_pcg_tea64", False))
        # order-independent (fixed-point) accumulation in the fused kernel: bit-reproducible renders (extension)
        self.deterministic = bool(props.get("amd_deterministic", False))
        # single-pass film lifecycle (extension): when one pass of the fused kernel renders all samples of all pixels, let its
        # row flush store the developed (H,W,T,3) tensor directly — no cleared 4-channel block, no develop pass.  Same values
        # (the weight channel is 0: develop divides by 1).  False keeps the reference's clear / accumulate / develop steps.
        self.direct_develop = bool(props.get("amd_direct_develop", True))
        # compute units left WITHOUT a workgroup of the persistent fused kernel (extension; mtr_render_params.reserve_cus): room for
        # the kernels of other streams — RCCL's film reduction of the previous row band — while a band renders.  0 on one GPU.
        self.reserve_cus = int(props.get("amd_reserve_cus", 0))
        self.mode = _cabi.MTR_MODE_AUTO            # kernel organisation (extension; not a reference key)
        m = props.get("amd_mode", None)
        if m is not None:
            self.mode = {"auto": 0, "fused": 1, "wavefront": 2}[m]
        self.max_wavefront_size = 2 ** 32          # common.py:51: lanes of one pass
        self.pass_wavefront_size = 2 ** 26 - 1     # common.py:60: lanes per pass once the render is split
        self.last_counters = None      # counters / kernel times of the last mtr_render call ...
        self.last_times = None
        self.total_counters = None     # ... and summed over every pass since the last prepare()
        self.total_times = None
        self.collect_stats = False

    @property
    def amd_mode(self):
        """kernel organisation (extension): "auto" | "fused" | "wavefront" """
        return {0: "auto", 1: "fused", 2: "wavefront"}[int(self.mode)]

    @amd_mode.setter
    def amd_mode(self, m):
        self.mode = {"auto": 0, "fused": 1, "wavefront": 2}[m]

    def aov_names(self):
        return []

    # -- common.py:32-85 ---------------------------------------------------
    def prepare(self, scene, sensor, seed, spp, aovs, _direct_develop=False):
        film = sensor.film()
        if hasattr(film, "direct_develop"):
            film.direct_develop = bool(_direct_develop)      # only render() asks for the single-pass lifecycle
        sampler = sensor.sampler().clone()
        if spp != 0:
            sampler.set_sample_count(spp)
        spp = sampler.sample_count()
        sampler.set_samples_per_wavefront(spp)
        film_size = film.crop_size()
        self.total_counters, self.total_times = None, None
        n_pixels = film_size[0] * film_size[1]
        wavefront_size = n_pixels * spp
        if wavefront_size > self.max_wavefront_size and int(self.pass_wavefront_size / n_pixels) == 0:
            raise Exception("Your film is too big. Please make it smaller.")      # (before the film's storage is allocated)
        film.prepare(aovs)
        if wavefront_size <= self.max_wavefront_size:
            sampler.seed(seed, wavefront_size)
            return [(sampler, spp)]
        # common.py:56-85: more than 2^32 samples cannot run in one pass (32-bit lane index); the reference goes down to
        # 2^26 per pass, each pass with its own sampler whose seed is drawn from a seeder sampler
        spp_per_pass = int(self.pass_wavefront_size / n_pixels)
        if spp_per_pass == 0:
            raise Exception("Your film is too big. Please make it smaller.")
        needs_remainder = spp % spp_per_pass != 0
        num_passes = spp // spp_per_pass + 1 * needs_remainder
        sampler.set_sample_count(num_passes)
        sampler.set_samples_per_wavefront(num_passes)
        sampler.seed(seed, num_passes)
        import numpy as np
        seeds = (sampler.next_1d_first() * np.float32(2 ** 32)).astype(np.uint32)      # mi.UInt32(sampler.next_1d() * 2**32)

        def sampler_per_pass(i):
            spp_i = spp % spp_per_pass if (needs_remainder and i == num_passes - 1) else spp_per_pass
            clone = sensor.sampler().clone()
            clone.set_sample_count(spp_i)
            clone.set_samples_per_wavefront(spp_i)
            clone.seed(int(seeds[i]), n_pixels * spp_i)
            return clone, spp_i

        return [sampler_per_pass(i) for i in range(num_passes)]

    def check_transient_(self, scene, sensor):
        if isinstance(sensor, int):
            sensor = scene.sensors()[sensor]
        if not isinstance(sensor.film(), TransientHDRFilm):
            raise AssertionError("The film of the sensor must be of type transient_hdr_film or phasor_hdr_film")

    def add_transient_f(self, film, pos, ray_weight, sample_scale):
        """common.py:411-422: closure that pre-multiplies the sample scale and splats."""
        return (lambda spec, distance, wavelengths=None, active=None, laser_x=None, laser_y=None:
                film.add_transient_data(pos, distance, wavelengths, spec * sample_scale, ray_weight, active))

    def _flags(self):
        f = 0
        if self.camera_unwarp:
            f |= _cabi.MTR_FLAG_CAMERA_UNWARP
        if self.discard_direct_light:
            f |= _cabi.MTR_FLAG_DISCARD_DIRECT_LIGHT
        if self.pcg_initseq_plus_lane:
            f |= _cabi.MTR_FLAG_PCG_INITSEQ_PLUS_LANE
        if self.pcg_tea64:
            f |= _cabi.MTR_FLAG_PCG_TEA64
        if self.deterministic:
            f |= _cabi.MTR_FLAG_DETERMINISTIC
        return f

    def render_params(self, film, seed_value, spp_total, spp_begin=0, spp_end=None,
                      pixel_begin=0, pixel_end=None, spp_scale=0) -> _cabi.mtr_render_params:
        p = _cabi.mtr_render_params()
        p.spp_total = spp_total
        p.spp_scale = spp_scale          # multi-pass renders: sample_scale = 1 / total_spp of all passes (common.py:173-175)
        p.spp_begin = spp_begin
        p.spp_end = spp_total if spp_end is None else spp_end
        cw, ch = film.crop_size()
        p.pixel_begin = pixel_begin
        p.pixel_end = cw * ch if pixel_end is None else pixel_end
        p.seed = seed_value & 0xFFFFFFFF
        p.max_depth = -1 if self.max_depth >= 0x7FFFFFFF else int(self.max_depth)
        p.rr_depth = int(self.rr_depth)
        p.flags = self._flags()
        p.mode = int(self.mode)
        p.reserve_cus = max(0, int(self.reserve_cus))
        return p

    # -- common.py:122-213 ---------------------------------------------------
    def render(self, scene, sensor=0, seed=0, spp=0, develop=True, evaluate=True,
               progress_callback=None, spp_range=None, pixel_range=None) -> Tuple[TensorXf, TensorXf]:
        if not develop:
            raise Exception("develop=True must be specified when invoking AD integrators")
        if isinstance(sensor, int):
            sensor = scene.sensors()[sensor]
        film = sensor.film()
        self.check_transient_(scene, sensor)
        direct = spp_range is None and pixel_range is None and self._direct_develop_ok(scene, sensor, spp)
        samplers_spps = self.prepare(scene=scene, sensor=sensor, seed=seed, spp=spp, aovs=self.aov_names(), _direct_develop=direct)
        total_spp = sum(s for _, s in samplers_spps)
        self.accumulate(scene, sensor, samplers_spps, total_spp, spp_range, pixel_range, progress_callback)
        return film.develop()

    def _direct_develop_ok(self, scene, sensor, spp):
        """one pass, every sample of every film pixel in one launch of the fused kernel with LDS rows: its row flush can
        store the developed tensor (MTR_FLAG_DEVELOPED_ROWS)"""
        film = sensor.film()
        if not self.direct_develop or type(film) is not TransientHDRFilm:
            return False
        if tuple(film.crop_size()) != tuple(film.size()) or tuple(film.crop_offset()) != (0, 0):
            return False                                    # rows outside the crop window would never be written
        spp = spp if spp != 0 else sensor.sampler().sample_count()
        W, H = film.size()
        if W * H * spp > self.max_wavefront_size:
            return False                                    # split render: several passes add into the block
        ctx = get_context()
        handle = scene.gpu_handle(ctx, sensor)
        params = self.render_params(film, 0, spp)
        mode, ok = C.c_uint32(0), C.c_uint32(0)
        ctx.check(ctx.lib.mtr_render_plan(handle, C.byref(params), C.byref(mode), C.byref(ok)), "mtr_render_plan")
        return bool(ok.value)

    def accumulate(self, scene, sensor, samplers_spps, total_spp, spp_range=None, pixel_range=None,
                   progress_callback=None, rows_are_zero=None, defer_stats=None, developed_partial=False, bands=None):
        """The pass loop of render() (common.py:157-210) without prepare/develop: ADDS into the film.
        ``rows_are_zero``: the caller vouches that the film rows of ``pixel_range`` are untouched since clear() (a render
        split into disjoint row bands: every band's FIRST pass may store its rows instead of read-modify-write).
        ``defer_stats``: "first" / "more" — the call stays asynchronous (no counters / timings are read back); the device
        counters are reset by "first" and keep summing through "more"; read them with ``fetch_counters()``.
        ``bands``: (n, epoch, device pointer of n uint32 words) — band completion words of ONE fused launch over the whole pixel
        range (mtr_render_params.n_bands): word b receives ``epoch`` when band b's rows are in the film."""
        film = sensor.film()
        ctx = get_context(film._device.index)
        ctx.bind_current_stream()
        handle = scene.gpu_handle(ctx, sensor)
        # ``developed_partial``: the caller wants the developed (H,W,T,3) rows of THIS call's samples and pixels only — partial
        # sums it reduces itself (multi-GPU: a 3-channel reduce-scatter instead of a 4-channel one, no clear, no develop)
        direct = film.developed_storage() if hasattr(film, "developed_storage") else None
        if direct is not None and (len(samplers_spps) > 1 or ((spp_range is not None or pixel_range is not None) and not developed_partial)):
            direct = None
            film._ensure_raw()                    # (a direct-develop film asked to accumulate in parts: back to the block)
        tptr = C.c_void_p((direct if direct is not None else film.transient_storage.torch_tensor()).data_ptr())
        sptr = C.c_void_p(film.steady_accum().data_ptr())
        multi = len(samplers_spps) > 1
        if multi and spp_range is not None:
            raise NotImplementedError("sample sharding of a multi-pass (> 2^32 lanes) render: shard by rows instead")
        for i, (sampler_i, spp_i) in enumerate(samplers_spps):
            s0, s1 = (0, spp_i) if spp_range is None else spp_range
            p0, p1 = (0, None) if pixel_range is None else pixel_range
            # a pass of a split render indexes its lanes with ITS sample count (its own sampler) and scales by the total
            params = self.render_params(film, sampler_i.seed_value(), spp_i if multi else total_spp, s0, s1, p0, p1,
                                        spp_scale=total_spp if multi else 0)
            if bands is not None:
                if multi:
                    raise NotImplementedError("band completion words of a multi-pass render")
                params.n_bands, params.band_epoch, params.band_done = int(bands[0]), int(bands[1]) & 0xFFFFFFFF, int(bands[2])
            if direct is not None:
                params.flags |= _cabi.MTR_FLAG_DEVELOPED_ROWS  # the row flush stores the developed (H,W,T,3) rows whole
            elif film.film_is_zero or (rows_are_zero and i == 0):
                params.flags |= _cabi.MTR_FLAG_FILM_ZERO      # first pass after clear(): row flushes may store
            film.film_is_zero = False
            stats_now = self.collect_stats and defer_stats is None
            if defer_stats == "more" or (defer_stats == "first" and i > 0):
                params.flags |= _cabi.MTR_FLAG_KEEP_COUNTERS
            cnt = _cabi.mtr_counters() if stats_now else None
            tim = _cabi.mtr_kernel_times() if stats_now else None
            ctx.check(ctx.lib.mtr_render(handle, C.byref(params), tptr, sptr,
                                         C.byref(cnt) if cnt is not None else None,
                                         C.byref(tim) if tim is not None else None), "mtr_render")
            if stats_now:
                self.last_counters, self.last_times = cnt.as_dict(), tim.as_dict()
                if self.total_counters is None:
                    self.total_counters = {k: 0 for k in self.last_counters if k != "reserved"}
                    self.total_times = {k: 0 for k in self.last_times}
                for k in self.total_counters:
                    self.total_counters[k] += self.last_counters[k]
                for k in self.total_times:
                    self.total_times[k] += self.last_times[k]
            if direct is not None:
                film._developed_written = True    # (a band of a banded render counts: its caller covers the other rows)
            if progress_callback:
                progress_callback((i + 1) / len(samplers_spps))

    def developed_rows_ok(self, scene, sensor, total_spp, spp_range=None, pixel_range=None):
        """would mtr_render honour MTR_FLAG_DEVELOPED_ROWS for this (partial) render of a plain transient_hdr_film?"""
        film = sensor.film()
        if not self.direct_develop or type(film) is not TransientHDRFilm:
            return False
        if tuple(film.crop_size()) != tuple(film.size()) or tuple(film.crop_offset()) != (0, 0):
            return False
        W, H = film.size()
        if W * H * total_spp > self.max_wavefront_size:
            return False
        ctx = get_context()
        handle = scene.gpu_handle(ctx, sensor)
        s0, s1 = (0, total_spp) if spp_range is None else spp_range
        p0, p1 = (0, None) if pixel_range is None else pixel_range
        params = self.render_params(film, 0, total_spp, s0, s1, p0, p1)
        mode, ok = C.c_uint32(0), C.c_uint32(0)
        ctx.check(ctx.lib.mtr_render_plan(handle, C.byref(params), C.byref(mode), C.byref(ok)), "mtr_render_plan")
        return bool(ok.value)

    def resolved_mode(self, scene, sensor, total_spp, spp_range=None, pixel_range=None):
        """the kernel organisation mtr_render would run (MTR_MODE_AUTO resolved by the library): "fused" | "wavefront".
        Callers that overlap several calls of one render on different streams need it: only the fused organisation
        keeps its per-launch state apart."""
        film = sensor.film()
        ctx = get_context(film._device.index)
        handle = scene.gpu_handle(ctx, sensor)
        s0, s1 = (0, total_spp) if spp_range is None else spp_range
        p0, p1 = (0, None) if pixel_range is None else pixel_range
        params = self.render_params(film, 0, total_spp, s0, s1, p0, p1)
        mode = C.c_uint32(0)
        ctx.check(ctx.lib.mtr_render_plan(handle, C.byref(params), C.byref(mode), None), "mtr_render_plan")
        return {_cabi.MTR_MODE_FUSED: "fused", _cabi.MTR_MODE_WAVEFRONT: "wavefront"}[int(mode.value)]

    def reset_counters(self, film):
        """zero the device counters on the CURRENT stream (then every call of the render passes defer_stats="more")"""
        ctx = get_context(film._device.index)
        ctx.bind_current_stream()
        ctx.check(ctx.lib.mtr_counters_reset(ctx.handle), "mtr_counters_reset")

    def fetch_counters(self, film):
        """counters summed on the device over the deferred calls of one render (the caller synchronised their streams)"""
        ctx = get_context(film._device.index)
        cnt = _cabi.mtr_counters()
        ctx.check(ctx.lib.mtr_counters_read(ctx.handle, C.byref(cnt)), "mtr_counters_read")
        self.last_counters = cnt.as_dict()
        self.total_counters = {k: v for k, v in self.last_counters.items() if k != "reserved"}
        return self.total_counters

    def render_forward(self, *a, **k):
        raise NotImplementedError("differentiable rendering (common.py:215-323) is outside the north-star path")

    def render_backward(self, *a, **k):
        raise NotImplementedError("differentiable rendering (common.py:325-409) is outside the north-star path")

    def to_string(self):
        return f"{type(self).__name__}[\n  max_depth = {self.max_depth}, \n  rr_depth = {self.rr_depth}\n]"

    __str__ = __repr__ = to_string
