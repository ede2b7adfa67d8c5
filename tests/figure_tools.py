"""Reading numbers back out of matplotlib figures (tests/golden/reference_figures.npz: figures embedded in the reference's
notebooks): the image area of an ``imshow``, its values through the inverse colour map and the colour bar's scale, and the
curve of a line plot.  Precision is that of a figure — a few percent — which is what the tests built on it claim."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def load_figures():
    z = np.load(os.path.join(HERE, "golden", "reference_figures.npz"))
    meta = json.loads(str(z["meta"]))
    return {k: z[k] for k in meta}, meta


def _runs(mask):
    """[(start, end)] of the True runs of a 1-D mask (end exclusive)"""
    m = np.concatenate([[False], mask, [False]])
    d = np.diff(m.astype(np.int8))
    return list(zip(np.where(d == 1)[0], np.where(d == -1)[0]))


def _dark(rgb):
    return rgb.astype(np.int32).sum(-1) < 120


def frames(rgb, x_from=0):
    """axes frames drawn with spines: [(y0, y1, x0, x1)] (spine pixel coordinates), left to right, right of ``x_from``"""
    dk = _dark(rgb)
    H, W = dk.shape
    cols = [c for c in range(x_from, W) if max((b - a for a, b in _runs(dk[:, c])), default=0) > 0.5 * H]
    col_runs = _runs(np.isin(np.arange(W), cols))
    xs = [(a + b - 1) // 2 for a, b in col_runs]             # spine centres
    out = []
    for k in range(0, len(xs) - 1, 2):
        x0, x1 = xs[k], xs[k + 1]
        rows = [r for r in range(H) if dk[r, x0:x1 + 1].mean() > 0.9]
        if rows:
            out.append((rows[0], rows[-1], x0, x1))
    return out


def hot_image_box(rgb):
    """``imshow(cmap='hot')`` with the axes off: the image is the big non-white block left of the colour bar"""
    nonwhite = rgb.astype(np.int32).sum(-1) < 740
    col_frac = nonwhite.mean(0)
    blocks = [(a, b) for a, b in _runs(col_frac > 0.5) if b - a > 100]
    x0, x1 = blocks[0]
    rows = _runs(nonwhite[:, x0:x1].mean(1) > 0.9)
    y0, y1 = max(rows, key=lambda r: r[1] - r[0])
    return y0, y1, x0, x1                                   # end exclusive


def cells(rgb, box, n_rows, n_cols, margin=0.25):
    """mean colour of the central part of each data cell of an n_rows x n_cols image drawn into ``box``"""
    y0, y1, x0, x1 = box
    out = np.zeros((n_rows, n_cols, 3), np.float64)
    ch, cw = (y1 - y0) / n_rows, (x1 - x0) / n_cols
    for i in range(n_rows):
        a, b = y0 + (i + margin) * ch, y0 + (i + 1 - margin) * ch
        ra, rb = int(np.floor(a)), max(int(np.ceil(b)), int(np.floor(a)) + 1)
        for j in range(n_cols):
            c, d = x0 + (j + margin) * cw, x0 + (j + 1 - margin) * cw
            ca, cb = int(np.floor(c)), max(int(np.ceil(d)), int(np.floor(c)) + 1)
            out[i, j] = rgb[ra:rb, ca:cb].reshape(-1, 3).mean(0)
    return out


def invert_cmap(colors, name):
    """colour -> position in [0, 1] of the matplotlib colour map ``name`` (nearest of 1024 samples)"""
    import matplotlib
    lut = matplotlib.colormaps[name](np.linspace(0.0, 1.0, 1024))[:, :3] * 255.0
    c = np.asarray(colors, np.float64).reshape(-1, 1, 3)
    idx = np.argmin(((c - lut[None]) ** 2).sum(-1), axis=1)
    return (idx / 1023.0).reshape(np.asarray(colors).shape[:-1])


def colorbar_range(rgb, tick_step, symmetric=False, x_from=0):
    """(vmin, vmax) of the colour bar: the bar is the right-most frame, its ticks are the dark marks just right of it;
    consecutive ticks are ``tick_step`` apart (read off the figure), the bottom of a 'hot' bar is 0 — a tick sits there —
    and a symmetric bar ('seismic', vmin = -vmax) has its 0 tick in the middle"""
    fr = frames(rgb, x_from)[-1]
    y0, y1, x0, x1 = fr
    dk = _dark(rgb)
    strip = dk[:, x1 + 2:x1 + 5].mean(1) > 0.6
    ticks = [0.5 * (a + b - 1) for a, b in _runs(strip) if b - a <= 3]
    d = np.median(np.diff(sorted(ticks)))
    span = (y1 - y0) / d * tick_step                       # value range covered by the bar
    return (-0.5 * span, 0.5 * span) if symmetric else (0.0, span)


def line_curve(rgb, meta):
    """the curve of a one-line plot: (x values, y values) per pixel column the line crosses (matplotlib's first colour)"""
    y0, y1, x0, x1 = frames(rgb)[0]
    dk = _dark(rgb)
    # tick marks: below the bottom spine / left of the left spine
    xt = [0.5 * (a + b - 1) for a, b in _runs(dk[y1 + 2:y1 + 5].mean(0) > 0.6) if b - a <= 3]
    yt = [0.5 * (a + b - 1) for a, b in _runs(dk[:, x0 - 4:x0 - 1].mean(1) > 0.6) if b - a <= 3]
    xt, yt = sorted(xt), sorted(yt, reverse=True)          # values grow to the right / upwards
    kx = meta["x_tick_step"] / np.median(np.diff(xt))
    ky = meta["y_tick_step"] / np.median(-np.diff(yt))
    blue = (np.abs(rgb.astype(np.int32) - np.array([31, 119, 180])).sum(-1) < 90)
    xs, ys = [], []
    for c in range(x0 + 1, x1):
        r = np.where(blue[y0 + 1:y1, c])[0]
        if len(r):
            xs.append(meta["x_first_tick"] + (c - xt[0]) * kx)
            ys.append(meta["y_first_tick"] + (yt[0] - (y0 + 1 + r.min())) * ky)      # upper envelope
    return np.asarray(xs), np.asarray(ys)


def ncc(a, b):
    """normalised cross-correlation of two arrays (scale- and offset-free agreement of structure)"""
    a = np.asarray(a, np.float64).ravel() - np.mean(a)
    b = np.asarray(b, np.float64).ravel() - np.mean(b)
    return float(a @ b / max(np.sqrt((a @ a) * (b @ b)), 1e-300))
