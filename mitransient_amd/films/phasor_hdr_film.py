"""``phasor_hdr_film`` plugin surface (mitransient/films/phasor_hdr_film.py): the transient film in the frequency
domain.  Same properties and defaults (:125-131), same choice of frequencies (:140-152, numpy ``fftfreq`` restricted
to the band of a Morlet wavelet ``wl_mean`` / ``wl_sigma``), same channel layout — raw ``(H, W, 2F+1)``: real and
imaginary part per frequency, then the weight (:171-186) — and ``develop()`` -> ``(steady, phasors (H, W, F, 2))``
(:210-238).  Monochromatic variants only, like the reference (:155-157).  The accumulation
``value * exp(i * fmod(-2 pi f (opl - start_opl), 2 pi))`` (render/phasor_image_block.py:42-67) runs in the HIP
library: the path kernels keep (opl, value) records per pixel and ``k_wf_phasor_scatter`` folds them per frequency.
"""
from __future__ import annotations

import ctypes as C
from typing import Sequence

import numpy as np

from .. import _cabi, variant
from ..runtime import get_context, require_gpu
from ..scene import Properties
from ..tensor import TensorXf
from .transient_hdr_film import TransientHDRFilm


class _Float(float):
    """a frequency as the reference exposes it: a one-lane Dr.Jit Float — the notebooks read it as ``f[0]``"""

    def __getitem__(self, i):
        if i not in (0, -1):
            raise IndexError(i)
        return float(self)

    def __len__(self):
        return 1


class PhasorImageBlock:
    """(H, W, 2F+1) float32 accumulator in HBM (render/phasor_image_block.py)."""

    def __init__(self, size, frequencies, device=None):
        self.size = tuple(int(v) for v in size)
        self.frequencies = frequencies
        self.channel_count = 2 * len(frequencies) + 1
        self._device = device
        self._tensor = None
        self.clear()

    @property
    def size_xyt(self):
        return (self.size[0], self.size[1], self.channel_count)

    def clear(self):
        torch = require_gpu()
        W, H = self.size
        dev = self._device if self._device is not None else torch.device("cuda", torch.cuda.current_device())
        if self._tensor is None or tuple(self._tensor.shape) != (H, W, self.channel_count):
            self._tensor = torch.zeros((H, W, self.channel_count), dtype=torch.float32, device=dev)
        else:
            self._tensor.zero_()

    def tensor(self):
        return TensorXf(self._tensor)

    def torch_tensor(self):
        return self._tensor


class PhasorHDRFilm(TransientHDRFilm):
    def __init__(self, props: Properties):
        super().__init__(props)
        self.wl_mean = float(props.get("wl_mean", 100.0))
        self.wl_sigma = float(props.get("wl_sigma", 1000.0))
        self.temporal_bins = int(props.get("temporal_bins", 4096))          # :128 (the transient film defaults to 2048)
        if self.crop_size_ != self.size_:
            raise ValueError("PhasorHDRFilm: crop_size must match size")
        if self.crop_offset_ != (0, 0):
            raise ValueError("PhasorHDRFilm: crop_offset must be (0, 0)")
        if self.sample_border_:
            raise ValueError("PhasorHDRFilm: sample_border must be False")
        if self.exhaustive_scan:
            raise ValueError("PhasorHDRFilm: exhaustive_scan is a transient_hdr_film property")
        # :140-152
        nt = self.temporal_bins
        mean_idx = (nt * self.bin_width_opl) / self.wl_mean
        sigma_idx = (nt * self.bin_width_opl) / (self.wl_sigma * 6)
        freq_min_idx = np.maximum(0, int(np.floor(mean_idx - 3 * sigma_idx)))
        freq_max_idx = np.minimum(nt // 2, int(np.ceil(mean_idx + 3 * sigma_idx)))
        self.frequencies_f32 = np.ascontiguousarray(
            np.fft.fftfreq(nt, d=self.bin_width_opl)[freq_min_idx:freq_max_idx + 1].astype(np.float32))
        if self.frequencies_f32.size == 0:
            raise ValueError("PhasorHDRFilm: wl_mean / wl_sigma select no frequency")
        self.frequencies = [_Float(f) for f in self.frequencies_f32]
        self.phasors = None

    # -- lifecycle -------------------------------------------------------------
    def raw_shape(self):
        W, H = self.size_
        return (H, W, 2 * len(self.frequencies) + 1)

    def prepare(self, aovs: Sequence[str] = ()):
        if not variant.is_monochromatic():
            raise RuntimeError("PhasorHDRFilm: Only monochromatic rendering supported")          # :155-157
        if aovs:
            raise RuntimeError("PhasorHDRFilm: AOVs not supported")
        return super().prepare(aovs)

    def prepare_transient_(self, aovs: Sequence[str] = ()):
        channels = []
        for j in range(len(self.frequencies)):                   # :176-182
            channels += [f"L_fq{j:03d}_Re", f"L_fq{j:03d}_Im"]
        channels.append("W")
        self.channels = channels
        self.crop_offset_xyt = (0, 0, 0)
        self.crop_size_xyt = (self.size_[0], self.size_[1], len(channels))
        if self.phasors is None or self.phasors.torch_tensor().device != self._device or \
                self.phasors.channel_count != len(channels):
            self.phasors = PhasorImageBlock(self.size_, self.frequencies, device=self._device)
        else:
            self.phasors.clear()
        self.transient_storage = self.phasors           # the integrator addresses the accumulator by this name
        self.film_is_zero = True
        return len(channels)

    def create_block(self):
        raise NotImplementedError("Not implemented for phasor_hdr_film")                          # :140-141

    def clear(self):
        if self._steady_accum is not None:
            self._steady_accum.zero_()
        if self.phasors is not None:
            self.phasors.clear()
            self.film_is_zero = True

    # -- splat from Python (:240-262) -----------------------------------------
    def add_transient_data(self, pos, distance, wavelengths, spec, ray_weight=1.0, active=None, **kwargs):
        """pos (n,2), distance (n,), spec (n,) or (n,k): channel 0 is used (monochromatic); already multiplied by the
        sample scale."""
        torch = require_gpu()
        dev = self._device
        pos = torch.as_tensor(pos, dtype=torch.float32, device=dev)
        distance = torch.as_tensor(distance, dtype=torch.float32, device=dev)
        spec = torch.as_tensor(spec, dtype=torch.float32, device=dev) * ray_weight
        if spec.dim() == 2:
            spec = spec[:, 0]
        W, H = self.size_
        px, py = torch.floor(pos[:, 0]).to(torch.int64), torch.floor(pos[:, 1]).to(torch.int64)
        ok = (px >= 0) & (px < W) & (py >= 0) & (py < H)
        if active is not None:
            ok &= torch.as_tensor(active, dtype=torch.bool, device=dev)
        pixel = torch.where(ok, py * W + px, torch.full_like(px, W * H)).to(torch.int32).contiguous()
        ctx = get_context(dev.index)
        ctx.bind_current_stream()
        arrs = [t.contiguous() for t in (distance, spec, spec, spec)]
        soa = _cabi.mtr_splat_soa(pixel.data_ptr(), arrs[0].data_ptr(), arrs[1].data_ptr(), arrs[2].data_ptr(),
                                  arrs[3].data_ptr(), int(pixel.numel()), None)
        fd = self.desc()
        ms = C.c_float(0)
        ctx.check(ctx.lib.mtr_splat_add(ctx.handle, C.byref(soa), C.byref(fd), 0,
                                        C.c_void_p(self.phasors.torch_tensor().data_ptr()), C.byref(ms)), "mtr_splat_add")
        self.film_is_zero = False
        return float(ms.value)

    # -- develop (:210-238) ------------------------------------------------------
    def develop(self, raw: bool = False):
        steady, _ = TransientHDRFilm._develop_steady(self, raw)
        return steady, self.develop_phasors_(raw=raw)

    def develop_transient_(self, raw: bool = False):
        return self.develop_phasors_(raw)

    def develop_phasors_(self, raw: bool = False):
        if self.phasors is None:
            raise RuntimeError("No phasor storage allocated, was prepare() called first?")
        if raw:
            return self.phasors.tensor()
        torch = require_gpu()
        data = self.phasors.torch_tensor()
        W, H = self.size_
        out = torch.empty((H, W, len(self.frequencies), 2), dtype=torch.float32, device=data.device)
        ctx = get_context(data.device.index)
        ctx.bind_current_stream()
        fd = self.desc()
        ctx.check(ctx.lib.mtr_film_develop(ctx.handle, C.byref(fd), C.c_void_p(data.data_ptr()),
                                           C.c_void_p(out.data_ptr()), None, None), "mtr_film_develop")
        return TensorXf(out)

    def develop_slab(self, raw_t, raw_s):
        raise NotImplementedError("multi-GPU row slabs are implemented for transient_hdr_film")

    def traverse(self, callback):
        callback.put("frequencies", self.frequencies, 0)
        callback.put("start_opl", self.start_opl, 0)

    def to_string(self):
        return (f"PhasorHDRFilm[\n  size = {self.size()},\n  frequencies = {self.frequencies},\n"
                f"  start_opl = {self.start_opl},\n]")

    __str__ = __repr__ = to_string
