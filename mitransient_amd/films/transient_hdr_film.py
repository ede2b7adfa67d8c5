"""``transient_hdr_film`` plugin surface (mitransient/films/transient_hdr_film.py).

Same property keys and defaults (:114-121), same channel layout — raw (H,W,T,4)
"RGBW" with W == 0 (:173-202), developed (H,W,T,3) (:220-248) — and the
same ``prepare`` / ``create_block`` / ``add_transient_data`` / ``develop`` / ``clear`` /
``traverse`` methods.  Storage is torch tensors in HBM; arithmetic is in the HIP library.
"""
from __future__ import annotations

import ctypes as C
from typing import Sequence

import numpy as np

from .. import _cabi, variant
from ..render.transient_image_block import TransientImageBlock
from ..runtime import get_context, require_gpu
from ..scene import Properties, film_desc_from
from ..tensor import TensorXf


class TransientHDRFilm:
    def __init__(self, props: Properties):
        # mi.Film base properties [mitsuba3: src/render/film.cpp]
        w, h = int(props.get("width", 768)), int(props.get("height", 576))
        self.size_ = (w, h)
        cw, ch = int(props.get("crop_width", w)), int(props.get("crop_height", h))
        cx, cy = int(props.get("crop_offset_x", 0)), int(props.get("crop_offset_y", 0))
        if cx < 0 or cy < 0 or cw <= 0 or ch <= 0 or cx + cw > w or cy + ch > h:
            raise ValueError("Invalid crop window specification!")
        self.crop_size_ = (cw, ch)
        self.crop_offset_ = (cx, cy)
        self.sample_border_ = bool(props.get("sample_border", False))
        rf = props.get("rfilter", None)
        self.rfilter_ = (rf.get("type") if isinstance(rf, dict) else rf) or "gaussian"   # mitsuba's default filter
        # transient_hdr_film.py:116-121
        self.temporal_bins = int(props.get("temporal_bins", 2048))
        self.bin_width_opl = float(props.get("bin_width_opl", 0.003))
        self.start_opl = float(props.get("start_opl", 0))
        self.exhaustive_scan = bool(props.get("exhaustive_scan", False))
        self.laser_scan_width = int(props.get("laser_scan_width", 0))
        self.laser_scan_height = int(props.get("laser_scan_height", 0))
        if self.exhaustive_scan and not (self.laser_scan_width > 0 and self.laser_scan_height > 0):
            raise ValueError("exhaustive_scan needs laser_scan_width and laser_scan_height > 0")
        self.channels = None
        self.transient_storage = None
        self._steady_accum = None     # (H, W, 4): sum of L, sample count
        self.film_is_zero = False     # True between clear()/prepare() and the first accumulated pass
        self._device = None
        # single-pass film lifecycle (extension): the integrator sets ``direct_develop`` before prepare() when ONE pass of
        # the fused kernel renders every sample of every pixel — its row flush then stores the DEVELOPED (H,W,T,3) tensor
        # whole (MTR_FLAG_DEVELOPED_ROWS), so the 4-channel block is neither allocated and cleared nor developed; the raw
        # block is rebuilt from it on request (its weight channel is identically 0)
        self.direct_develop = False
        self._developed = None
        self._developed_written = False   # a render has stored every row of _developed (before that it is torch.empty)
        self._developed_given = False     # develop() handed _developed to the caller: the next prepare() must not reuse it

    # -- mi.Film accessors -------------------------------------------------
    def size(self):
        return self.size_

    def crop_size(self):
        return self.crop_size_

    def crop_offset(self):
        return self.crop_offset_

    def sample_border(self):
        return self.sample_border_

    def rfilter(self):
        return self.rfilter_

    def end_opl(self):
        return self.start_opl + self.bin_width_opl * self.temporal_bins

    def base_channels_count(self):
        return 3

    def desc(self) -> _cabi.mtr_film_desc:
        return film_desc_from(self)

    # -- lifecycle -----------------------------------------------------------
    def prepare(self, aovs: Sequence[str] = ()):
        if aovs:
            raise NotImplementedError("AOVs are not part of the transient_path hot path")
        torch = require_gpu()
        W, H = self.size_
        dev = torch.device("cuda", torch.cuda.current_device())
        self._device = dev
        # steady `hdrfilm` accumulator (transient_hdr_film.py:131-144): rgb + weight
        if self._steady_accum is None or tuple(self._steady_accum.shape) != (H, W, 4) or self._steady_accum.device != dev:
            self._steady_accum = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
        else:
            self._steady_accum.zero_()
        return self.prepare_transient_(aovs)

    def prepare_transient_(self, aovs: Sequence[str] = ()):
        # Film base flags carry no Alpha -> "RGBW"; "LW" in the monochromatic variants (:177-179).  The accumulator in HBM is
        # RGBW either way: mono renders run three equal channels (variant.py) and the outputs below keep channel 0
        self.channels = list("LW" if variant.is_monochromatic() else "RGBW") + list(aovs)
        self.crop_offset_xyt = (self.crop_offset_[0], self.crop_offset_[1], 0)
        self.crop_size_xyt = (self.size_[0], self.size_[1], self.temporal_bins)
        if self.direct_develop:
            torch = require_gpu()
            shape = self.raw_shape()[:-1] + (3,)
            # develop() returns this tensor ITSELF (no 3 GiB copy), so once it has been handed out it belongs to the caller — the
            # reference returns a fresh tensor from every develop() — and the next render gets a new one (the caching allocator
            # hands back the block of a result the caller has dropped; one the caller still holds stays valid)
            if (self._developed is None or self._developed_given or tuple(self._developed.shape) != shape
                    or self._developed.device != self._device):
                self._developed = None            # (free before allocating: two 3 GiB tensors need not coexist)
                self._developed = torch.empty(shape, dtype=torch.float32, device=self._device)
            self._developed_written = False
            self._developed_given = False
            self.transient_storage = None         # the raw block is not needed (and its 4 GiB are not held)
            self.film_is_zero = False
            return len(self.channels)
        self._developed = None
        self._developed_written = self._developed_given = False
        self.transient_storage = self.create_block()
        self.film_is_zero = True
        return len(self.channels)

    def developed_storage(self):
        """the (H, W, T, 3) tensor a direct-develop render writes (None in the ordinary lifecycle)"""
        return self._developed if self.direct_develop else None

    def _ensure_raw(self):
        """the 4-channel accumulator, rebuilt from a direct-develop render when something needs it (raw output, more
        contributions from Python, a second pass): raw = (developed rgb, weight 0)"""
        if self.transient_storage is None and self._developed is not None:
            self.transient_storage = self.create_block()              # zero-filled
            if self._developed_written:
                self.transient_storage.torch_tensor()[..., :3].copy_(self._developed)
            else:
                # nothing has been rendered into the direct tensor yet (it is torch.empty): the block starts from zero, as
                # after prepare() in the ordinary lifecycle, and the first pass may store its rows
                self.film_is_zero = True
            self._developed = None
            self._developed_written = self._developed_given = False
            self.direct_develop = False
        return self.transient_storage

    def create_block(self):
        n_ch = 4                                            # storage channels (see prepare_transient_)
        if (self.transient_storage is not None and self.transient_storage.size_xyt == self.crop_size_xyt
                and self.transient_storage.torch_tensor().device == self._device
                and tuple(self.transient_storage.torch_tensor().shape) == self.raw_shape()):
            self.transient_storage.clear()                  # reuse the allocation: same zero-filled state
            return self.transient_storage
        return TransientImageBlock(size_xyt=self.crop_size_xyt, offset_xyt=self.crop_offset_xyt,
                                   exhaustive_scan=self.exhaustive_scan, laser_scan_width=self.laser_scan_width,
                                   laser_scan_height=self.laser_scan_height,
                                   channel_count=n_ch, rfilter=self.rfilter_, device=self._device)

    def raw_shape(self):
        W, H = self.size_
        if self.exhaustive_scan:
            return (H, W, self.laser_scan_height, self.laser_scan_width, self.temporal_bins, 4)
        return (H, W, self.temporal_bins, 4)

    def clear(self):
        if self._steady_accum is not None:
            self._steady_accum.zero_()
        if self.transient_storage is None and self._developed is not None:
            self._developed = None
            self._developed_written = self._developed_given = False
            self.direct_develop = False
            self.transient_storage = self.create_block()
            self.film_is_zero = True
        elif self.transient_storage:
            self.transient_storage.clear()
            self.film_is_zero = True

    def steady_accum(self):
        return self._steady_accum

    # -- splat from Python (transient_hdr_film.py:250-276) -----------------
    def add_transient_data(self, pos, distance, wavelengths, spec, ray_weight=1.0, active=None,
                           laser_x=0, laser_y=0, variant=1):
        """pos: (n,2) pixel coordinates (incl. crop offset); distance: (n,); spec: (n,3) already
        multiplied by the sample scale.  Device torch tensors (or anything torch.as_tensor takes).
        ``variant`` (extension): 1 (default) = the library looks whether the contributions come pixel by pixel — the order
        of the reference's own lanes, pixel * spp + s — and then adds each pixel's run through an LDS row (31 % of the HBM
        roofline); any other order falls back, on the device, to variant 0 = one f32 atomic per channel (the contract form,
        2 %)."""
        torch = require_gpu()
        dev = self._device
        pos = torch.as_tensor(pos, dtype=torch.float32, device=dev)
        distance = torch.as_tensor(distance, dtype=torch.float32, device=dev)
        spec = torch.as_tensor(spec, dtype=torch.float32, device=dev) * ray_weight
        if spec.dim() == 1 or spec.shape[1] == 1:           # monochromatic: one value per contribution
            spec = spec.reshape(-1, 1).expand(-1, 3)
        px = torch.floor(pos[:, 0]).to(torch.int64) - self.crop_offset_[0]
        py = torch.floor(pos[:, 1]).to(torch.int64) - self.crop_offset_[1]
        W, H = self.size_
        ok = (px >= 0) & (px < W) & (py >= 0) & (py < H)
        if active is not None:
            ok &= torch.as_tensor(active, dtype=torch.bool, device=dev)
        pixel = torch.where(ok, py * W + px, torch.full_like(px, W * H))   # out-of-range id -> dropped by the kernel
        self._ensure_raw()
        self.film_is_zero = False                 # (the block is zero-filled either way; splats add with atomics / RMW)
        laser = None
        if self.exhaustive_scan:            # row position laser_x * Lh + laser_y (transient_image_block.py:136-138)
            lx = torch.as_tensor(laser_x, dtype=torch.int64, device=dev).expand(px.shape)
            ly = torch.as_tensor(laser_y, dtype=torch.int64, device=dev).expand(px.shape)
            lok = (lx >= 0) & (lx < self.laser_scan_width) & (ly >= 0) & (ly < self.laser_scan_height)
            laser = torch.where(lok, lx * self.laser_scan_height + ly,
                                torch.full_like(lx, self.laser_scan_width * self.laser_scan_height))
        return self.transient_storage.put_opl(pixel, distance, spec[:, 0], spec[:, 1], spec[:, 2], self.desc(), variant,
                                              laser=laser)

    # -- develop -------------------------------------------------------------
    def develop(self, raw: bool = False):
        transient_image = self.develop_transient_(raw=raw)
        if self.exhaustive_scan:            # transient_hdr_film.py:213-214: dr.mean(transient_image, axis=-1)
            return TensorXf(transient_image.torch().mean(dim=-1)), transient_image
        return self._develop_steady(raw)[0], transient_image

    def _develop_steady(self, raw: bool = False):
        """``self.steady.develop(raw)`` of the reference: (H,W,3) (one channel in the monochromatic variants), or the
        raw (H,W,4) accumulator"""
        torch = require_gpu()
        W, H = self.size_
        transient_image = None
        ctx = get_context(self._device.index)
        ctx.bind_current_stream()
        if raw:
            steady = self._steady_accum
        else:
            steady = torch.empty((H, W, 3), dtype=torch.float32, device=self._device)
            fd = self.desc()
            ctx.check(ctx.lib.mtr_film_develop(ctx.handle, C.byref(fd), None, None,
                                               C.c_void_p(self._steady_accum.data_ptr()),
                                               C.c_void_p(steady.data_ptr())), "mtr_film_develop")
            if variant.is_monochromatic():
                steady = steady[..., :1].contiguous()        # pixel_format 'luminance' (:141)
        # the reference builds its steady hdrfilm WITH the crop window (:131-144), so steady.develop() is
        # (crop_h, crop_w, C); only the transient block spans the full size (crop_size_xyt = size, :185-187).  The
        # accumulator here is full-size with the window at the top-left corner (positions carry the crop offset).
        cw, ch = self.crop_size_
        if (cw, ch) != (W, H):
            steady = steady[:ch, :cw].contiguous()
        return TensorXf(steady), transient_image

    def develop_transient_(self, raw: bool = False):
        torch = require_gpu()
        if self._developed is not None and self.transient_storage is None:
            if raw:
                self._ensure_raw()
            else:
                if not self._developed_written:
                    raise RuntimeError("develop() before any render wrote the film")
                out = self._developed             # written whole by the render: already developed
                self._developed_given = True      # the caller owns it from here on (prepare_transient_ allocates anew)
                return TensorXf(out[..., :1].contiguous() if variant.is_monochromatic() else out)
        if not self.transient_storage:
            raise RuntimeError("No transient storage allocated, was prepare_transient_() called first?")
        if raw and variant.is_monochromatic():
            t = self.transient_storage.torch_tensor()
            return TensorXf(torch.stack((t[..., 0], t[..., 3]), dim=-1))          # "LW"
        if raw:
            return self.transient_storage.tensor
        W, H = self.size_
        data = self.transient_storage.torch_tensor()
        out = torch.empty(self.raw_shape()[:-1] + (3,), dtype=torch.float32, device=data.device)
        ctx = get_context(data.device.index)
        ctx.bind_current_stream()
        fd = self.desc()
        ctx.check(ctx.lib.mtr_film_develop(ctx.handle, C.byref(fd), C.c_void_p(data.data_ptr()),
                                           C.c_void_p(out.data_ptr()), None, None), "mtr_film_develop")
        if variant.is_monochromatic():
            out = out[..., :1].contiguous()
        return TensorXf(out)

    def develop_slab(self, raw_t, raw_s, out=None):
        """develop() of a row slab (rows, W, T, 4) / (rows, W, 4): used by the multi-GPU path after the
        reduce-scatter, so that every GPU develops 1/N of the film.  ``out``: (transient, steady) contiguous
        destination tensors of the slab's developed shape (written in place), or None to allocate."""
        torch = require_gpu()
        if raw_t is None:                         # the steady image alone (the transient rows came developed out of the render)
            rows = int(raw_s.shape[0])
            out_s = torch.empty((rows, self.size_[0], 3), dtype=torch.float32, device=raw_s.device) if out is None else out
            if rows:
                ctx = get_context(raw_s.device.index)
                ctx.bind_current_stream()
                fd = self.desc()
                fd.height = rows
                fd.crop_height = min(fd.crop_height, rows)
                fd.crop_offset_y = 0
                raw_s = raw_s.contiguous()
                ctx.check(ctx.lib.mtr_film_develop(ctx.handle, C.byref(fd), None, None, C.c_void_p(raw_s.data_ptr()),
                                                   C.c_void_p(out_s.data_ptr())), "mtr_film_develop")
            return None, out_s
        rows = int(raw_t.shape[0])
        W = self.size_[0]
        if out is not None:
            out_t, out_s = out
            assert out_t.is_contiguous() and out_s.is_contiguous() and out_t.shape[0] == rows and out_s.shape[0] == rows
        else:
            out_t = torch.empty((rows,) + self.raw_shape()[1:-1] + (3,), dtype=torch.float32, device=raw_t.device)
            out_s = torch.empty((rows, W, 3), dtype=torch.float32, device=raw_t.device)
        if rows == 0:
            return out_t, out_s
        ctx = get_context(raw_t.device.index)
        ctx.bind_current_stream()
        fd = self.desc()
        fd.height = rows
        fd.crop_height = min(fd.crop_height, rows)
        fd.crop_offset_y = 0
        raw_t, raw_s = raw_t.contiguous(), raw_s.contiguous()
        ctx.check(ctx.lib.mtr_film_develop(ctx.handle, C.byref(fd), C.c_void_p(raw_t.data_ptr()),
                                           C.c_void_p(out_t.data_ptr()), C.c_void_p(raw_s.data_ptr()),
                                           C.c_void_p(out_s.data_ptr())), "mtr_film_develop")
        return out_t, out_s

    # -- introspection ---------------------------------------------------------
    def traverse(self, callback):
        for k in ("temporal_bins", "bin_width_opl", "start_opl", "exhaustive_scan",
                  "laser_scan_width", "laser_scan_height"):
            callback.put(k, getattr(self, k), 0)

    def parameters_changed(self, keys=()):
        pass

    def to_string(self):
        return ("TransientHDRFilm[\n"
                f"  exhaustive_scan = {self.exhaustive_scan},\n  size = {self.size()},\n"
                f"  crop_size = {self.crop_size()},\n  crop_offset = {self.crop_offset()},\n"
                f"  sample_border = {self.sample_border()},\n  filter = {self.rfilter()},\n"
                f"  temporal_bins = {self.temporal_bins},\n  bin_width_opl = {self.bin_width_opl},\n"
                f"  start_opl = {self.start_opl},\n]")

    __str__ = __repr__ = to_string
