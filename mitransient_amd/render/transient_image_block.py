"""``TransientImageBlock``: the (H, W, T, C) float32 time-resolved accumulator in HBM.

Mirrors mitransient/render/transient_image_block.py: ``clear`` (:56-70) zero-fills
the tensor, ``put`` / ``put_`` / ``accum`` (:79-151) are the box-filter splat whose
arithmetic — bin = floor((opl - start)/width), flat index ((y*W + x)*T + bin)*C + k,
C float adds — runs in the HIP scatter-add kernels (``mtr_splat_add`` when called
from Python with arrays; fused into the path kernels during ``render``).
"""
from __future__ import annotations

import ctypes as C

from .. import _cabi
from ..runtime import get_context, require_gpu
from ..tensor import TensorXf


class TransientImageBlock:
    def __init__(self, size_xyt, offset_xyt=(0, 0, 0), exhaustive_scan=False,
                 laser_scan_width=0, laser_scan_height=0, channel_count=4, rfilter=None,
                 border=False, warn_negative=False, warn_invalid=False, device=None):
        if exhaustive_scan and not (int(laser_scan_width) > 0 and int(laser_scan_height) > 0):
            raise ValueError("exhaustive_scan needs laser_scan_width and laser_scan_height > 0")
        if rfilter is not None and rfilter != "box":
            # transient_image_block.py:150-151
            raise RuntimeError("TransientImageBlock::put_(): using a rfilter but it is not supported. "
                               "If you need this, please open an issue on GitHub.")
        if channel_count != 4:
            raise NotImplementedError("only the RGBW (C=4) channel layout of the rgb variant is supported")
        self.offset_xyt = tuple(int(v) for v in offset_xyt)
        self.size_xyt = tuple(int(v) for v in size_xyt)
        self.exhaustive_scan = bool(exhaustive_scan)
        self.laser_scan_width = int(laser_scan_width) if exhaustive_scan else 0
        self.laser_scan_height = int(laser_scan_height) if exhaustive_scan else 0
        self.channel_count = channel_count
        self.rfilter = None
        self.border_size = 0
        self.warn_negative = warn_negative
        self.warn_invalid = warn_invalid
        self._device = device
        self._tensor = None
        self.clear()

    # -- storage ----------------------------------------------------------
    def clear(self):
        torch = require_gpu()
        W, H, T = self.size_xyt
        dev = self._device if self._device is not None else torch.device("cuda", torch.cuda.current_device())
        # transient_image_block.py:63-68: (H, W, laser_scan_height, laser_scan_width, T, C) when exhaustive
        shape = ((H, W, self.laser_scan_height, self.laser_scan_width, T, self.channel_count) if self.exhaustive_scan
                 else (H, W, T, self.channel_count))
        if self._tensor is None or tuple(self._tensor.shape) != shape:
            self._tensor = torch.empty(shape, dtype=torch.float32, device=dev)
        # TransientImageBlock.clear (transient_image_block.py:56-70) = mtr_film_clear on the context's stream
        fd = _cabi.mtr_film_desc()
        fd.width, fd.height, fd.crop_width, fd.crop_height = W, H, W, H
        fd.temporal_bins, fd.bin_width_opl = T, 1.0
        fd.laser_scan_width, fd.laser_scan_height = self.laser_scan_width, self.laser_scan_height
        ctx = get_context(self._tensor.device.index)
        ctx.bind_current_stream()
        ctx.check(ctx.lib.mtr_film_clear(ctx.handle, C.byref(fd), C.c_void_p(self._tensor.data_ptr()), None), "mtr_film_clear")

    @property
    def tensor(self) -> TensorXf:
        return TensorXf(self._tensor)

    def torch_tensor(self):
        return self._tensor

    def set_size(self, size_xyt):
        size_xyt = tuple(int(v) for v in size_xyt)
        if size_xyt != self.size_xyt:
            self.size_xyt = size_xyt

    # -- splatting --------------------------------------------------------
    def put_opl(self, pixel, opl, r, g, b, film_desc: _cabi.mtr_film_desc, variant: int = 0, laser=None):
        """Scatter-add n time-resolved contributions (device torch tensors): pixel u32 (y*W+x),
        opl f32, r/g/b f32, laser u32 (laser_x*laser_scan_height + laser_y; exhaustive films).  This is
        add_transient_data + put_ + accum in one HIP launch."""
        torch = require_gpu()
        ctx = get_context(self._tensor.device.index)
        ctx.bind_current_stream()
        n = int(pixel.numel())
        pix = pixel.to(device=self._tensor.device, dtype=torch.int32).contiguous()
        arrs = [t.to(device=self._tensor.device, dtype=torch.float32).contiguous() for t in (opl, r, g, b)]
        if self.warn_negative or self.warn_invalid:
            # transient_image_block.py:107-125: the checked values are the channels [r, g, b, alpha = 0, weight = 0]
            valid = torch.ones_like(arrs[1], dtype=torch.bool)
            for v in arrs[1:]:
                if self.warn_negative:
                    valid &= v >= -1e-5
                if self.warn_invalid:
                    valid &= torch.isfinite(v)
            npix = self.size_xyt[0] * self.size_xyt[1]
            bad = (~valid) & (pix.to(torch.int64) & 0xFFFFFFFF < npix)          # `active` lanes only
            if bool(bad.any()):
                k = int(torch.nonzero(bad)[0])
                import logging
                logging.getLogger("mitransient_amd").warning(
                    "Invalid sample value: [%s, %s, %s, 0.0, 0.0]", float(arrs[1][k]), float(arrs[2][k]), float(arrs[3][k]))
        las = laser.to(device=self._tensor.device, dtype=torch.int32).contiguous() if laser is not None else None
        soa = _cabi.mtr_splat_soa(pix.data_ptr(), arrs[0].data_ptr(), arrs[1].data_ptr(),
                                  arrs[2].data_ptr(), arrs[3].data_ptr(), n, las.data_ptr() if las is not None else None)
        ms = C.c_float(0)
        ctx.check(ctx.lib.mtr_splat_add(ctx.handle, C.byref(soa), C.byref(film_desc), int(variant),
                                        C.c_void_p(self._tensor.data_ptr()), C.byref(ms)), "mtr_splat_add")
        return float(ms.value)

    def to_string(self):
        return (f"{type(self).__name__}[\n  offset_xyt = {self.offset_xyt}"
                f"  size_xyt = {self.size_xyt}, \n  channel_count = {self.channel_count}, \n"
                f"  border_size = {self.border_size}, \n  rfilter = BoxFilter[] \n]")

    __str__ = __repr__ = to_string
