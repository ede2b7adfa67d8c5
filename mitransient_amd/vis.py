"""``mitransient.vis`` for the unpolarized variants (mitransient/unpolarized_visualization.py): host-side numpy /
matplotlib helpers the reference's notebooks call — ``tonemap_transient`` (:14-18), ``tonemap_grad_transient`` (:21-39),
``save_video`` (:42-62), ``save_frames`` (:65-77), ``show_video`` (:80-122), ``rainbow_visualization`` (:125-151).
Same names, arguments and results; nothing here touches the GPU."""
from __future__ import annotations

import os
import struct

import numpy as np


def _frame(a, axis, i):
    return np.take(a, i, axis=axis)


def to_srgb_uint8(img, srgb_gamma: bool = True):
    """linear float image -> uint8, with the sRGB transfer function (what ``mi.Bitmap.convert(UInt8, srgb_gamma=True)``
    and ``mi.util.convert_to_bitmap`` do)"""
    x = np.clip(np.nan_to_num(np.asarray(img, dtype=np.float64)), 0.0, 1.0)
    if srgb_gamma:
        x = np.where(x <= 0.0031308, 12.92 * x, 1.055 * np.power(x, 1.0 / 2.4) - 0.055)
    out = np.rint(x * 255.0).astype(np.uint8)
    if out.ndim == 3 and out.shape[-1] == 1:
        out = np.repeat(out, 3, axis=-1)
    return out


def tonemap_transient(transient, scaling=1.0):
    """linear tonemap: divide by the 99th percentile of |transient|"""
    tnp = np.array(transient)
    return tnp / np.quantile(np.abs(tnp), 0.99) * scaling


def tonemap_grad_transient(transient, axis_video=2):
    """gradient video -> 'coolwarm' colours: values are normalised by the 99.9th percentile of their magnitude,
    clipped to [-1, 1] and looked up at (v + 1) / 2"""
    if axis_video != 2:
        raise AssertionError("axis_video must be 2")
    tnp = np.array(transient)
    if tnp.ndim == 4:
        tnp = tnp.mean(axis=-1)
    import matplotlib
    cmap = matplotlib.colormaps["coolwarm"] if hasattr(matplotlib, "colormaps") else __import__("matplotlib.cm").cm.get_cmap("coolwarm")
    v = np.clip(tnp.astype(np.float32) / np.float32(np.quantile(np.abs(tnp), 0.999)), -1.0, 1.0)
    return cmap((v + 1.0) / 2.0)[..., :3].astype(np.float32)


def save_video(path, transient, axis_video=2, fps=24, display_video=False):
    """the transient image as an .mp4 (OpenCV 'mp4v'), one frame per index of ``axis_video``"""
    try:
        import cv2
    except ImportError as e:                               # same hard dependency as the reference
        raise ImportError("save_video needs OpenCV (cv2)") from e
    transient = np.asarray(transient)
    h, w = _frame(transient, axis_video, 0).shape[:2]
    writer = cv2.VideoWriter(path, cv2.VideoWriter_fourcc(*"mp4v"), fps, (w, h))
    for i in range(transient.shape[axis_video]):
        writer.write(to_srgb_uint8(_frame(transient, axis_video, i))[:, :, ::-1])      # RGB -> BGR
    writer.release()
    if display_video:
        from IPython.display import Video, display
        return display(Video(path, embed=True, width=w, height=h))


def write_exr(path, img):
    """minimal OpenEXR 2 writer: scan lines, no compression, 32-bit float channels (alphabetical order, as the format
    requires); enough for ``save_frames`` without an EXR library"""
    img = np.asarray(img, dtype=np.float32)
    if img.ndim == 2:
        img = img[..., None]
    h, w, c = img.shape
    names = {1: ["Y"], 3: ["B", "G", "R"], 4: ["A", "B", "G", "R"]}.get(c)
    if names is None:
        raise ValueError("write_exr: 1, 3 or 4 channels")
    order = {"Y": 0, "R": 0, "G": 1, "B": 2, "A": 3}

    def attr(name, typ, payload):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(payload)) + payload

    chlist = b"".join(n.encode() + b"\0" + struct.pack("<iBBBBii", 2, 0, 0, 0, 0, 1, 1) for n in names) + b"\0"
    box = struct.pack("<iiii", 0, 0, w - 1, h - 1)
    header = (attr("channels", "chlist", chlist) + attr("compression", "compression", b"\0") + attr("dataWindow", "box2i", box) +
              attr("displayWindow", "box2i", box) + attr("lineOrder", "lineOrder", b"\0") +
              attr("pixelAspectRatio", "float", struct.pack("<f", 1.0)) + attr("screenWindowCenter", "v2f", struct.pack("<ff", 0.0, 0.0)) +
              attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0")
    line_bytes = c * w * 4
    offset0 = 8 + len(header) + 8 * h
    with open(path, "wb") as fh:
        fh.write(struct.pack("<ii", 20000630, 2) + header)
        fh.write(b"".join(struct.pack("<Q", offset0 + y * (8 + line_bytes)) for y in range(h)))
        for y in range(h):
            fh.write(struct.pack("<ii", y, line_bytes))
            for n in names:
                fh.write(np.ascontiguousarray(img[y, :, order[n]]).tobytes())


def save_frames(data, folder, axis_video=2):
    """one ``NNN.exr`` per frame of ``data`` along ``axis_video``"""
    os.makedirs(folder, exist_ok=True)
    data = np.asarray(data)
    for i in range(data.shape[axis_video]):
        write_exr(os.path.join(folder, f"{i:03d}.exr"), _frame(data, axis_video, i))


def show_video(input_sample, axis_video=2, uint8_srgb=True, normalize=False):
    """plays the transient video inline (IPython / Jupyter): a matplotlib animation rendered to HTML5 video"""
    import matplotlib.animation as animation
    from IPython.display import HTML, display
    from matplotlib import pyplot as plt
    data = np.asarray(input_sample)
    peak = data.max()

    def picture(i):
        fr = _frame(data, axis_video, i)
        if normalize:
            fr = fr / peak
        return to_srgb_uint8(fr, srgb_gamma=True) if uint8_srgb else np.clip(fr, 0.0, 1.0)

    fig = plt.figure()
    im = plt.imshow(picture(0))
    plt.axis("off")

    def update(i):
        im.set_data(picture(i))
        return im

    ani = animation.FuncAnimation(fig, update, frames=data.shape[axis_video], repeat=False)
    display(HTML(ani.to_html5_video()))
    plt.close()


def rainbow_visualization(steady_state, data_transient, modulo, min_modulo, max_modulo,
                          max_time_bins=None, mode="peak_time_fusion", scale_fusion=1):
    """time-of-flight as colour (Jarabo 2012): each pixel's peak time bin picks a 'jet' colour; pixels whose peak bin
    falls in [min_modulo, max_modulo] modulo ``modulo`` form iso-time bands.  ``sparse_fusion``: bands show the steady
    image; ``rainbow_fusion``: bands show the colour; ``peak_time_fusion``: colour on the bands, steady image elsewhere."""
    import matplotlib
    jet = matplotlib.colormaps["jet"] if hasattr(matplotlib, "colormaps") else __import__("matplotlib.cm").cm.jet
    steady_state = np.asarray(steady_state)
    data_transient = np.asarray(data_transient)
    n_bins = data_transient.shape[2] if max_time_bins is None else max_time_bins
    peak_bin = data_transient.max(axis=-1).argmax(axis=-1)
    phase = peak_bin % modulo
    band = (phase >= min_modulo) & (phase <= max_modulo)
    colour = jet(peak_bin / n_bins)[..., :3]
    out = np.zeros_like(steady_state)
    if mode == "sparse_fusion":
        out[band] = steady_state[band] ** scale_fusion
    elif mode == "rainbow_fusion":
        out[band] = colour[band]
    elif mode == "peak_time_fusion":
        out[band] = colour[band]
        out[~band] = steady_state[~band] ** scale_fusion
    else:
        raise NotImplementedError("Mode not implemented")
    return out
