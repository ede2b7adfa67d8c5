"""The renderer against figures the REFERENCE ITSELF produced: the PNGs embedded in the notebooks under
/root/reference/examples (real mitransient 1.2.0 on Mitsuba 3.6.4 / 3.7.0), decoded into tests/golden/reference_figures.npz by
tests/golden/make_reference_figures.py.  Values are read back through the inverse colour map and the colour bar's ticks
(tests/figure_tools.py), so the precision is that of a figure: a few percent on sums, a bin or two on the time axis.  That is
enough to catch what every oracle-vs-kernel test shares with the oracle — flipped axes, a shifted time origin (the near-clip
offset of the optical path), unit and normalisation errors, the sign convention of the phasor film, the luminance of the
monochromatic variants — and these figures are the only reference-produced data there is (its own tests assert shapes).

CPU tests render with the ORACLE at a few thousand samples; the ``gpu`` tests render the same scenes with the product at the
notebooks' sample counts (and the exhaustive capture, which is too large for the CPU suite).  Parity stays "unpinned" in the
strict sense — no sample-for-sample comparison with Mitsuba exists — but no longer unchecked.
"""
import numpy as np
import pytest

import figure_tools as ft

matplotlib = pytest.importorskip("matplotlib")


@pytest.fixture(scope="module")
def figures():
    return ft.load_figures()


# ---------------------------------------------------------------------------------------------------------------------
# scenes of the notebooks
def nlos_notebook_scene(capture, spp, res=64, exhaustive=False):
    """examples/transient-nlos/1-simple-nlos-scenes.ipynb cells 3-5, 17, 23 (llvm_ad_mono): Z.obj at z = 1, relay rectangle
    with a nlos_capture_meter, projector irradiance 100 / fov 0.2 at (-0.5, 0, 0.25), 300 bins of 0.006 from 1.85,
    integrator defaults (max_depth 6, rr_depth 5), laser focused on pixel (32, 32) for the single capture"""
    import mitransient_amd.mi as mi
    from mitransient_amd.scenes import nlos_z
    mi.set_variant("llvm_ad_mono")
    extra = dict(exhaustive_scan=True, laser_scan_width=res, laser_scan_height=res) if exhaustive else None
    integ = dict(max_depth=6, rr_depth=5, nlos_hidden_geometry_sampling_do_rroulette=False)
    if exhaustive:
        integ.update(nlos_hidden_geometry_sampling_includes_relay_wall=False, discard_direct_paths=False)
    return nlos_z(width=res, height=res, temporal_bins=300, bin_width_opl=0.006, start_opl=1.85, capture=capture, spp=spp,
                  irradiance=100.0, film_extra=extra, **integ)


def cbox_scene(spp, freq=False):
    """examples/transient/cornell-box/cbox_diffuse.xml (400 x 400, 400 bins of 6.5 from 1000, max_depth 8) — or
    cbox_diffuse_freq.xml (llvm_ad_mono, phasor_hdr_film 200 x 200, wl_mean 100, wl_sigma 100, 4000 bins of 1 from 0, max_depth 5,
    discard_direct_light) — from the data fixture tests/golden/cbox_diffuse_scene.npz (the XML's flattened geometry)"""
    import os
    import mitransient_amd.mi as mi
    from mitransient_amd.scenes import from_fixture
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cbox_diffuse_scene.npz")
    if not freq:
        mi.set_variant("llvm_ad_rgb")
        return from_fixture(path, spp=spp)
    mi.set_variant("llvm_ad_mono")
    return from_fixture(path, spp=spp,
                        film={"type": "phasor_hdr_film", "width": 200, "height": 200, "wl_mean": 100.0, "wl_sigma": 100.0,
                              "temporal_bins": 4000, "bin_width_opl": 1.0, "start_opl": 0.0},
                        integrator={"max_depth": 5, "discard_direct_light": True})


def nlos_camera_scene(name, spp, tmp_path):
    """examples/transient-nlos/nlos-z-simple.xml / nlos-z-room.xml (2-complex-nlos-scenes.ipynb, llvm_ad_mono): transient_nlos_path
    behind a perspective camera, projector at the camera's pose, `ply` planes + the hidden Z — from the data fixture
    tests/golden/nlos_xml_scenes.npz (the XML in dictionary form, the meshes as triangles, written back as .obj files)"""
    import json
    import os
    import mitransient_amd.mi as mi
    from mitransient_amd.scenes import _unjson
    mi.set_variant("llvm_ad_mono")
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "nlos_xml_scenes.npz"))
    d = _unjson(json.loads(str(z[f"{name}/dict"])))
    for key, v in d.items():
        if isinstance(v, dict) and v.get("type") == "ply":
            tris = z[v["filename"]]
            path = os.path.join(str(tmp_path), f"{name}_{key}.obj")
            with open(path, "w") as fh:
                fh.write("\n".join([f"v {float(a)!r} {float(b)!r} {float(c)!r}" for a, b, c in tris.reshape(-1, 3)] +
                                   [f"f {3 * i + 1} {3 * i + 2} {3 * i + 3}" for i in range(len(tris))]) + "\n")
            v.update(type="obj", filename=path, face_normals=True)       # (the NLOS tier shades flat)
        if isinstance(v, dict) and v.get("type") == "perspective":
            v["sampler"]["sample_count"] = spp
    return mi.load_dict(d)


# per figure: the least correlation a correct render reaches (the 25 000-spp figures of the dim late frames are noisy themselves)
CAMERA_NCC = {"nlos_cam_simple_t130": 0.95, "nlos_cam_simple_t140": 0.9, "nlos_cam_simple_t150": 0.88,
              "nlos_cam_room_t130": 0.88, "nlos_cam_room_t140": 0.75, "nlos_cam_room_t150": 0.7, "nlos_cam_room_t110": 0.75,
              "nlos_cam_room_t120": 0.75, "nlos_cam_room_t200": 0.55, "nlos_cam_room_t210": 0.5}


def check_nlos_camera_frames(figures, name, tr):
    """2-complex-nlos-scenes.ipynb, cells 5 / 11 / 13: frames of the 32 x 32 x 300 tensor.  Structure, the left-right orientation
    (the scenes are symmetric top to bottom), and the absolute scale — values of 1e-5 / 1e-6 here, five orders of magnitude
    below the capture-meter scenes of notebook 1: the camera's pixel measure instead of the relay wall's.  Every frame sum of
    BOTH scenes within 8 % (measured: nlos-z-room 0.98 .. 1.06, nlos-z-simple 0.987 .. 1.000).  Round 3 had nlos-z-simple at a
    uniform 0.886 .. 0.897: its laser points at the CENTRE of a two-triangle wall, i.e. at the triangles' shared diagonal, and
    one connection ray in nine of emitter_laser_targets_sample fell between them (a Moller-Trumbore crack; closed since round
    4: mtr_core.h kEdgeEps, tests/test_scene_host.py::test_shared_edges_are_closed); the room's wall is centred elsewhere."""
    figs, meta = figures
    for fig, m in meta.items():
        if m.get("scene") != name:
            continue
        box = ft.hot_image_box(figs[fig])
        _, hi = ft.colorbar_range(figs[fig], m["tick_step"], x_from=box[3] + 3)
        ref = ft.invert_cmap(ft.cells(figs[fig], box, 32, 32), "hot") * hi
        mine = tr[:, :, m["t"]]
        c = ft.ncc(ref, mine)
        assert c >= CAMERA_NCC[fig], (fig, c)
        assert ft.ncc(ref, mine[:, ::-1]) <= c - 0.15 and ft.ncc(ref, mine.T) <= c - 0.3, fig
        assert abs(mine.sum() / ref.sum() - 1.0) <= 0.08, (fig, mine.sum(), ref.sum())


def oracle_render(oracle, scene, spp, seeds=(0,)):
    """developed transient tensor (and steady image) of the ORACLE, averaged over independent seeds"""
    sd = scene.data()
    integ, film = scene.integrator(), scene.sensors()[0].film()
    acc_t = acc_s = None
    for seed in seeds:
        t4, s4, _ = oracle.render(sd, integ.render_params(film, seed, spp))
        acc_t = t4 if acc_t is None else acc_t + t4
        acc_s = s4 if acc_s is None else acc_s + s4
    return acc_t / len(seeds), acc_s


def product_render(scene, spp):
    import torch
    steady, transient = scene.integrator().render(scene, seed=0, spp=spp)
    torch.cuda.synchronize()
    return np.array(transient), np.array(steady)


# ---------------------------------------------------------------------------------------------------------------------
# comparisons (shared by the CPU and the GPU tests)
def check_nlos_frames(figures, capture, frames_of, ncc_min, sum_tol):
    """frames_of(t) -> (res, res) image of time bin t.  Structure (normalised cross-correlation), orientation (a mirrored or
    transposed frame must agree clearly worse), the time axis (two bins earlier / later agree worse on the sharp frames) and
    the ABSOLUTE scale (frame sum against the colour bar's scale: irradiance, pixel weights, 1 / spp)."""
    figs, meta = figures
    for name, m in meta.items():
        if m["kind"] != "hot" or m["capture"] != capture:
            continue
        box = ft.hot_image_box(figs[name])
        _, hi = ft.colorbar_range(figs[name], m["tick_step"], x_from=box[3] + 3)
        ref = ft.invert_cmap(ft.cells(figs[name], box, 64, 64), "hot") * hi
        mine = frames_of(m["t"])
        c = ft.ncc(ref, mine)
        assert c >= ncc_min, (name, c)
        assert max(ft.ncc(ref, mine[:, ::-1]), ft.ncc(ref, mine[::-1]), ft.ncc(ref, mine.T)) <= c - 0.02, name
        if name in ("nlos_single_t30", "nlos_confocal_t23"):              # thin wave fronts: two bins off is visibly off
            assert max(ft.ncc(ref, frames_of(m["t"] + 2)), ft.ncc(ref, frames_of(m["t"] - 2))) <= c - 0.15, name
        assert abs(mine.sum() / ref.sum() - 1.0) <= sum_tol, (name, mine.sum(), ref.sum())


def check_nlos_pixel_curve(figures, curve, peak_tol):
    """np.array(data_transient)[11, 11, :, 0] of the single capture: first arrival, peak position and height, the second
    lobe, the end of the signal"""
    figs, meta = figures
    m = meta["nlos_single_pixel_11_11"]
    x, y = ft.line_curve(figs["nlos_single_pixel_11_11"], m)
    ref = np.interp(np.arange(300), x, y)
    assert abs(float(x[np.argmax(y)]) - float(np.argmax(curve))) <= 1.5                  # peak: bin 55 - 56
    assert abs(float(x[np.argmax(y > 0.003)]) - float(np.argmax(curve > 0.003))) <= 2.0   # first arrival
    assert abs(curve.max() / y.max() - 1.0) <= peak_tol, (curve.max(), y.max())
    assert abs(curve[78:92].mean() / ref[78:92].mean() - 1.0) <= 0.12                    # second lobe (about 0.022)
    last = 299 - int(np.argmax(curve[::-1] > 0.0008))
    assert 155 <= last <= 178, last                                                      # figure: about 162 - 172
    assert np.all(curve[:50] == 0.0) and ft.ncc(ref, curve) >= 0.85


def check_cbox_rainbow(figures, t3):
    """cell 10 of 4-rainbow_visualization.ipynb: mode='rainbow_fusion', modulo 20, bands = peak bin mod 20 in [0, 5], coloured
    jet(peak_bin / 200).  The figure's band pixels must sit on this render's bands, with the phase centred where the figure's
    is (a shifted time origin — e.g. the near-clip distance counted into the optical path: 10 units = 1.5 bins — moves it),
    its black pixels off them, and the colours must name the same bins."""
    from scipy.ndimage import binary_erosion
    figs, meta = figures
    y0, y1, x0, x1 = ft.frames(figs["cbox_rainbow_fusion"])[0]
    rgb = figs["cbox_rainbow_fusion"][y0 + 1:y1, x0 + 1:x1].astype(np.float64)
    H, W = rgb.shape[:2]
    lut = matplotlib.colormaps["jet"](np.linspace(0, 1, 1024))[:, :3] * 255.0
    d = ((rgb.reshape(-1, 1, 3) - lut[None]) ** 2).sum(-1)
    idx, dist = d.argmin(1), np.sqrt(d.min(1))
    band = ((rgb.sum(-1).ravel() > 150) & (dist < 12)).reshape(H, W)          # a pure jet colour, not a blend with black
    black = (rgb.sum(-1) < 30)
    band_in, black_in = binary_erosion(band, iterations=2), binary_erosion(black, iterations=2)
    yy, xx = np.mgrid[0:H, 0:W]
    my, mx = np.clip(((yy + 0.5) * 400 / H).astype(int), 0, 399), np.clip(((xx + 0.5) * 400 / W).astype(int), 0, 399)
    peak = t3.max(axis=-1).argmax(axis=-1)
    results = {}
    for label, pk in (("as is", peak[my, mx]), ("flip lr", peak[my, 399 - mx]), ("flip ud", peak[399 - my, mx])):
        ph = pk % 20
        on_band = np.isin(ph[band_in], (19, 0, 1, 2, 3, 4, 5, 6)).mean()
        ang = 2 * np.pi * ph[band_in] / 20.0
        centre = (np.angle(np.exp(1j * ang).mean()) % (2 * np.pi)) * 20.0 / (2 * np.pi)      # circular mean of the phase
        results[label] = (on_band, centre, (ph[black_in] <= 5).mean())
    on_band, centre, on_black = results["as is"]
    assert on_band >= 0.7, results
    assert 1.7 <= centre <= 3.3, results                      # the band is bins 0 .. 5: centre 2.5
    assert on_black <= 0.2, results                           # (30 % if the two were unrelated)
    # (the room itself is left-right symmetric: only the two boxes tell the mirror image apart)
    assert results["flip lr"][0] <= on_band - 0.08 and results["flip ud"][0] <= on_band - 0.3, results
    dl = (peak[my, mx] - idx.reshape(H, W) / 1023.0 * 200.0)[band_in]
    assert abs(np.median(dl)) <= 2.0, np.median(dl)            # absolute bin named by the colour


def check_cbox_steady(figures, s3):
    """cell 11: mode='sparse_fusion' shows (steady ** 0.8) / max on the bands (peak bin mod 10 in [0, 3]) — the notebook's
    tonemapped steady image, i.e. the Cornell box's colours as real Mitsuba rendered them"""
    from scipy.ndimage import binary_erosion
    figs, meta = figures
    y0, y1, x0, x1 = ft.frames(figs["cbox_rainbow_fusion"])[0]               # same canvas layout for the three figures
    rgb = figs["cbox_sparse_fusion"][y0 + 1:y1, x0 + 1:x1].astype(np.float64) / 255.0
    H, W = rgb.shape[:2]
    steady = np.array((s3 / np.quantile(s3, 0.99)) ** (1.0 / 2.2))
    steady[steady > 1] = 1
    yy, xx = np.mgrid[0:H, 0:W]
    my, mx = np.clip(((yy + 0.5) * 400 / H).astype(int), 0, 399), np.clip(((xx + 0.5) * 400 / W).astype(int), 0, 399)
    mine_full = steady[my, mx] ** 0.8
    # the figure's lit pixels, away from band edges (where the canvas resampling blends them with black); which pixels are
    # lit is the figure's business (its own peak bins, noisy on the walls) — only their colour is compared
    lit = binary_erosion(rgb.sum(-1) > 0.1, iterations=1) & (rgb.max(-1) < 0.98)          # (the light itself is clipped)
    assert lit.sum() > 10000
    k = np.median(rgb[lit].sum(-1) / np.maximum(mine_full[lit].sum(-1), 1e-6))            # the figure is divided by its maximum
    mine_full = mine_full * k
    # structure: both images averaged over the lit pixels of 5 x 5 neighbourhoods (the render here has 1 / 85 of the figure's samples)
    from scipy.ndimage import uniform_filter
    w = lit.astype(np.float64)
    den = uniform_filter(w, 5)
    ok = den > 0.3

    def smooth(img):
        return np.stack([uniform_filter(img[..., c] * w, 5) for c in range(3)], -1)[ok] / den[ok][:, None]
    c = ft.ncc(smooth(rgb), smooth(mine_full))
    assert c >= 0.9 and ft.ncc(smooth(rgb), smooth(mine_full[:, ::-1])) <= c - 0.2, c
    regions = {"floor": (slice(300, 345), slice(90, 200)), "back wall": (slice(90, 140), slice(110, 230)),
               "left (red) wall": (slice(140, 240), slice(8, 40)), "right (green) wall": (slice(140, 240), slice(330, 362)),
               "tall box front": (slice(180, 300), slice(115, 215))}
    # per surface: the colour (chromaticity to 0.02) and, more loosely, the brightness — the canvas resampling blends band
    # edges with black, the deeper inside the bands the closer the figure's values come (3-pixel erosion: 0.90 .. 0.99 of the
    # render's on thin .. wide bands), so the level is held to 15 %
    deep = binary_erosion(rgb.sum(-1) > 0.1, iterations=3) & (rgb.max(-1) < 0.98)
    for label, sl in regions.items():
        sel = deep[sl]
        assert sel.sum() >= 100, label
        a, b = rgb[sl][sel].mean(0), (mine_full / k)[sl][sel].mean(0)            # (the light is on a band: the figure's maximum is 1)
        assert np.all(np.abs(a / a.sum() - b / b.sum()) <= 0.02), (label, a, b)
        assert np.all(np.abs(a - b) <= 0.03 + 0.15 * a), (label, a, b)
        assert np.argmax(a) == np.argmax(b), (label, a, b)                       # the surface's dominant colour


def check_cbox_freq(figures, phasors, ncc_min):
    """cell 14 of 3-frequency_space_rendering.ipynb: Re of frequency i = 0, 10 .. 40, 'seismic' between -max|data| and
    +max|data| (both parts, all 41 frequencies).  Sign and phase convention of the phasor film, the frequency list, the
    monochromatic variant's luminance, the scale."""
    figs, meta = figures
    F = phasors.shape[2]
    for name, m in meta.items():
        if m["kind"] != "seismic":
            continue
        fr = ft.frames(figs[name])[0]
        _, hi = ft.colorbar_range(figs[name], m["tick_step"], symmetric=True)
        ref = (ft.invert_cmap(ft.cells(figs[name], (fr[0] + 1, fr[1], fr[2] + 1, fr[3]), 200, 200, margin=0.1), "seismic") * 2 - 1) * hi
        i = m["freq_index"]
        re, im = phasors[:, :, i, 0], phasors[:, :, i, 1]
        c = ft.ncc(ref, re)
        assert c >= ncc_min[i], (name, c)
        assert abs(ft.ncc(ref, im)) <= 0.3 and ft.ncc(ref, phasors[:, :, (i + 3) % F, 0]) <= c - 0.3, name
        assert abs(np.sqrt((re ** 2).mean()) / np.sqrt((ref ** 2).mean()) - 1.0) <= 0.25, name
        # the colour bar's limit is the largest |value| of the whole tensor
        assert abs(np.abs(phasors).max() / hi - 1.0) <= 0.15, (np.abs(phasors).max(), hi)


# ---------------------------------------------------------------------------------------------------------------------
# CPU: the oracle
# ---------------------------------------------------------------------------------------------------------------------
# the README's images of the reference's own renders (/root/reference/.images): steady Cornell box, steady staircase
LUMA = np.array([0.2126, 0.7152, 0.0722])


def display(steady):
    """what a viewer shows of a linear image: clipped, gamma 2.2 (the curve the two README images are consistent with)"""
    return np.clip(np.asarray(steady, np.float64), 0.0, 1.0) ** (1.0 / 2.2)


def figure_content(rgb, bright_margin):
    """the image inside the figure's margin (white for the staircase; white, then a black frame, for the Cornell box)"""
    a = rgb.astype(np.float64) / 255.0 if rgb.dtype == np.uint8 else rgb
    edge = (a.min(-1) > 0.97) if bright_margin else (a.sum(-1) < 0.15)
    rows, cols = np.where(edge.mean(1) < 0.5)[0], np.where(edge.mean(0) < 0.5)[0]
    return a[rows[0]:rows[-1] + 1, cols[0]:cols[-1] + 1]


def resized(img, shape):
    from PIL import Image
    u8 = np.rint(np.clip(img, 0, 1) * 255).astype(np.uint8)
    return np.asarray(Image.fromarray(u8).resize((shape[1], shape[0]), Image.BOX if shape[0] < img.shape[0] else Image.BILINEAR)).astype(np.float64) / 255.0


def best_shift(ref, mine, reach=4, margin=8):
    best = (-2.0, 0, 0)
    for dy in range(-reach, reach + 1):
        for dx in range(-reach, reach + 1):
            c = ft.ncc(ref[margin:-margin, margin:-margin], np.roll(np.roll(mine, dy, 0), dx, 1)[margin:-margin, margin:-margin])
            best = max(best, (c, dy, dx))
    return best


def check_readme_staircase(figures, steady, ncc_min, interior_min, mean_tol):
    """`.images/staircase_steady.png` (README.md:26) against a steady render of BASELINE config 5's scene AS ITS FILE DESCRIBES
    IT (GGX lobes, vertex normals, bitmap textures): the same picture — structure (luminance correlation, best alignment at
    zero shift, mirrored hypotheses clearly worse) and RADIOMETRY: the mean linear colour of the whole image, which the camera,
    every material, the textures and the light's radiance enter (measured on the GPU: (0.239, 0.146, 0.080) against the
    figure's (0.239, 0.144, 0.081))"""
    fig = figures[0]["readme_staircase_steady"]
    ref = figure_content(fig, bright_margin=True)
    assert abs(ref.shape[1] / ref.shape[0] - 9.0 / 16.0) < 0.01
    small = steady.shape[0] < ref.shape[0]
    a, b = (resized(ref, steady.shape[:2]), display(steady)) if small else (ref, resized(display(steady), ref.shape[:2]))
    la, lb = a @ LUMA, b @ LUMA
    c = ft.ncc(la, lb)
    assert c >= ncc_min, c
    assert ft.ncc(la, lb[:, ::-1]) <= c - 0.12 and ft.ncc(la, lb[::-1]) <= c - 0.5
    m = 8 if not small else 3
    ci, dy, dx = best_shift(la, lb, reach=3, margin=m)
    assert (dy, dx) == (0, 0) and ci >= interior_min, (ci, dy, dx)
    mean_ref, mean_mine = (a ** 2.2).reshape(-1, 3).mean(0), (b ** 2.2).reshape(-1, 3).mean(0)
    assert np.all(np.abs(mean_mine / mean_ref - 1.0) <= mean_tol), (mean_ref, mean_mine)
    return c, mean_mine / mean_ref


# regions of the Cornell box image in units of its side: (y0, y1, x0, x1)
CBOX_REGIONS = {"left wall": (0.31, 0.70, 0.03, 0.125), "right wall": (0.31, 0.70, 0.875, 0.97), "back wall": (0.31, 0.39, 0.31, 0.70),
                "floor": (0.92, 0.985, 0.23, 0.47), "ceiling": (0.025, 0.08, 0.16, 0.31), "tall box front": (0.55, 0.78, 0.33, 0.47)}
# ... of which these lie BEHIND the luminaire's front edge (z > 238 of the box's 559): their level is held tightly (see below)
CBOX_INTERIOR = ("back wall", "tall box front")


def srgb_to_linear(x):
    x = np.asarray(x, np.float64)
    return np.where(x <= 0.04045, x / 12.92, ((x + 0.055) / 1.055) ** 2.4)


def linear_to_srgb(x):
    x = np.clip(np.asarray(x, np.float64), 0.0, 1.0)
    return np.where(x <= 0.0031308, 12.92 * x, 1.055 * x ** (1.0 / 2.4) - 0.055)


def resized_linear(img, n):
    """float image -> n x n, channel by channel (box filter going down, bilinear going up): averages LINEAR values"""
    from PIL import Image
    how = Image.BOX if n < img.shape[0] else Image.BILINEAR
    return np.stack([np.asarray(Image.fromarray(np.ascontiguousarray(img[..., c], np.float32), mode="F").resize((n, n), how)) for c in range(3)],
                    -1).astype(np.float64)


def check_readme_cornell_box(figures, steady, ncc_min, interior_min, chroma_tol, level_tol):
    """`.images/cornell-box.png` (README.md:20, docs/index.rst:1).  WHICH scene it shows was the open question of round 3 (blue
    0.3 - 0.4 x, green 0.85 x of a render of mitransient.cornell_box()).  Answered in round 4: it is a render of
    examples/transient/cornell-box/cbox_diffuse.xml — the light's radiance (18.387, 10.9873, 2.75357) of :59 instead of
    utils.py:156-160's (18.387, 13.9873, 6.75357), GRAY boxes (0.85, :41-43) instead of white ones — written through the sRGB
    transfer curve (not a 2.2 power: only with the true curve do values of 0.0015 and of 0.38 agree at once).  Against a
    render of that file every surface reads the SAME ratio in red, green and blue — e.g. left wall 1.096 / 1.083 / 1.092,
    tall box front 0.991 / 1.007 / 1.003, blue spanning 0.0015 ... 0.035 — asserted to `chroma_tol`; the LEVEL is 1.00 - 1.02 on
    the surfaces behind the luminaire's front edge (`level_tol`) and an achromatic 1.08 ... 1.21 on the parts of walls, floor
    and ceiling in front of it (z < 238 of 559), which the reference's own notebook render of the same file
    (4-rainbow_visualization.ipynb, `cbox_sparse_fusion`: no such ring against this render) does not show: an exposure
    gradient of that one image, recorded in HISTORY.md (part 2, section 2) and bounded here.  A render of mitransient.cornell_box() fails
    this check (asserted by the callers)."""
    fig = figures[0]["readme_cornell_box"]
    ref = figure_content(figure_content(fig, bright_margin=True), bright_margin=False)      # white margin, then the black frame
    assert abs(ref.shape[0] - ref.shape[1]) <= 4
    n = steady.shape[0]
    lin = np.asarray(steady, np.float64)
    fig_lin = resized_linear(srgb_to_linear(ref), n)
    a, b = linear_to_srgb(fig_lin), linear_to_srgb(lin)
    for ch, least in ((0, ncc_min), (1, ncc_min), (2, ncc_min - 0.08)):
        c = ft.ncc(a[..., ch], b[..., ch])
        assert c >= least, (ch, c)
        assert max(ft.ncc(a[..., ch], b[:, ::-1, ch]), ft.ncc(a[..., ch], b[::-1, :, ch]), ft.ncc(a[..., ch], b[..., ch].T)) <= c - 0.15, ch
    ci, dy, dx = best_shift(a[..., 0], b[..., 0], reach=3, margin=max(4, n // 16))
    # (at the figure's own resolution one pixel — 0.3 % of the side — is the precision of cropping its 356 x 359 content out of
    # the black frame)
    assert max(abs(dy), abs(dx)) <= (1 if n > 200 else 0) and ci >= interior_min, (ci, dy, dx)
    ratios = {}
    for name, (y0, y1, x0, x1) in CBOX_REGIONS.items():
        sl = (slice(int(y0 * n), int(y1 * n)), slice(int(x0 * n), int(x1 * n)))
        f, m = fig_lin[sl].reshape(-1, 3).mean(0), lin[sl].reshape(-1, 3).mean(0)
        r = ratios[name] = f / m
        # the COLOUR of every surface, in all three channels (blue included: it is 1 / 100 of red on the left wall)
        assert r.max() / r.min() - 1.0 <= chroma_tol, (name, r, f, m)
        if name in CBOX_INTERIOR:
            assert np.all(np.abs(r - 1.0) <= level_tol), (name, r)
        else:
            assert 1.0 - level_tol <= r.mean() <= 1.27, (name, r)
    # colours name the walls: red left, green right, the rest warm white
    assert a[int(0.5 * n), int(0.07 * n), 0] > 3 * a[int(0.5 * n), int(0.07 * n), 1]
    assert a[int(0.5 * n), int(0.93 * n), 1] > a[int(0.5 * n), int(0.93 * n), 0] and b[int(0.5 * n), int(0.93 * n), 1] > b[int(0.5 * n), int(0.93 * n), 0]
    return ratios


def check_readme_staircase_video(figures, transient, least_ncc, min_exact, last_exact=66, late_tol=3):
    """`.images/staircase_transient.gif` (README.md:27; 279 frames) against the transient tensor of the scene file's own film
    (400 bins of 0.1 from OPL 0, camera_unwarp): every frame of the video is found among the time bins by normalised
    cross-correlation of the tone-mapped frames (mitr.vis.tonemap_transient: value / quantile(|value|, 0.99), clipped) — and
    frame k is bin k + 20: exactly, in every one of the frames 6 ... 66 (the GPU render; first light to the wavefront reaching the
    floor), and within 3 bins up to k = 110, where the room has filled with light and consecutive frames are hard to tell apart.  Start of the time axis, bin width, camera_unwarp and the speed of the
    wavefront through config 5's scene: an offset, a slope or an unwarp error of ONE bin (0.1 units) would show"""
    frames, meta = figures[0]["readme_staircase_transient"], figures[1]["readme_staircase_transient"]
    H, W = frames.shape[1:3]
    t = np.asarray(transient, np.float64)
    fy, fx = t.shape[0] // H, t.shape[1] // W
    t = t[:H * fy, :W * fx].reshape(H, fy, W, fx, t.shape[2], 3).mean((1, 3))
    video = np.clip(t / np.quantile(np.abs(t), 0.99), 0.0, 1.0).sum(-1)            # (H, W, T)
    v = video - video.mean((0, 1), keepdims=True)
    vn = np.sqrt((v * v).sum((0, 1)))
    ks, bins, scores = [], [], []
    for k, f in zip(meta["frames"], frames.astype(np.float64).sum(-1) / 255.0):
        if not 6 <= k <= 110:
            continue
        g = f - f.mean()
        c = np.einsum("hw,hwt->t", g, v) / (np.sqrt((g * g).sum()) * np.where(vn > 0, vn, 1.0))
        ks.append(k); bins.append(int(np.argmax(c))); scores.append(float(c.max()))
    ks, bins, scores = np.array(ks), np.array(bins), np.array(scores)
    off = bins - ks
    early = ks <= last_exact         # the wavefront is still crossing the room: consecutive frames differ visibly
    if late_tol is None:             # (a noisy render: the late, slowly changing frames cannot be told apart)
        ks, bins, scores, off = ks[early], bins[early], scores[early], off[early]
        early = early[early]
    assert np.median(off) == 20 and np.mean(off[early] == 20) >= min_exact and np.all(np.abs(off[early] - 20) <= 1), list(zip(ks, bins))
    assert np.all(np.abs(off[~early] - 20) <= (late_tol or 0)), list(zip(ks, bins))
    slope, intercept = np.polyfit(ks[early], bins[early], 1)
    assert abs(slope - 1.0) <= 0.01 and abs(intercept - 20.0) <= 0.5, (slope, intercept)
    assert scores.min() >= least_ncc and np.median(scores) >= least_ncc + 0.05, (scores.min(), np.median(scores))
    return off, scores


def readme_staircase_scene(width, height, spp, **kw):
    from mitransient_amd.scenes import staircase
    return staircase(width=width, height=height, spp=spp, temporal_bins=kw.pop("temporal_bins", 64), materials="rough", vertex_normals=True,
                     textures=True, **kw)


def readme_cornell_scene(n, spp, which="xml"):
    """the scene of `.images/cornell-box.png`: examples/transient/cornell-box/cbox_diffuse.xml (data fixture
    tests/golden/cbox_diffuse_scene.npz) at n x n pixels — or, `which="dict"`, mitransient.cornell_box() (BASELINE configs 1-3),
    which the image is NOT (other radiance, white boxes)"""
    import mitransient_amd as mitr
    import mitransient_amd.mi as mi
    mi.set_variant("llvm_ad_rgb")
    if which == "xml":
        import os
        from mitransient_amd.scenes import from_fixture
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cbox_diffuse_scene.npz")
        return from_fixture(path, spp=spp, film={"width": n, "height": n, "temporal_bins": 16, "bin_width_opl": 200.0})
    d = mitr.cornell_box()
    d["sensor"]["film"].update(width=n, height=n, temporal_bins=16)
    d["sensor"]["sampler"]["sample_count"] = spp
    return mi.load_dict(d)


def test_oracle_matches_the_readme_images(oracle, figures):
    """the ORACLE against the two images of its own renders the reference shows in its README"""
    for which in ("xml", "dict"):
        scene = readme_cornell_scene(128, 512, which)
        sd, film = scene.data(), scene.sensors()[0].film()
        t4, s4, _ = oracle.render(sd, scene.integrator().render_params(film, 0, 512), use_bvh=True)
        _, s3 = oracle.develop(sd.film, t4, s4)
        if which == "xml":
            check_readme_cornell_box(figures, s3, ncc_min=0.8, interior_min=0.9, chroma_tol=0.05, level_tol=0.06)
        else:               # mitransient.cornell_box() is another picture: its light is 1.27 x greener and 2.45 x bluer, its boxes are white
            with pytest.raises(AssertionError):
                check_readme_cornell_box(figures, s3, ncc_min=0.8, interior_min=0.9, chroma_tol=0.05, level_tol=0.06)
    scene = readme_staircase_scene(54, 96, 160, temporal_bins=400)        # the scene file's own film: 400 bins of 0.1 from 0
    sd, film = scene.data(), scene.sensors()[0].film()
    assert (film.temporal_bins, film.bin_width_opl, film.start_opl, scene.integrator().camera_unwarp) == (400, 0.1, 0.0, True)
    t4, s4, _ = oracle.render(sd, scene.integrator().render_params(film, 0, 160), use_bvh=True)
    t3, s3 = oracle.develop(sd.film, t4, s4)
    check_readme_staircase(figures, s3, ncc_min=0.95, interior_min=0.95, mean_tol=0.12)
    check_readme_staircase_video(figures, t3, least_ncc=0.6, min_exact=0.8, last_exact=64, late_tol=None)


@pytest.mark.parametrize("capture", ["single", "confocal"])
def test_oracle_nlos_frames_match_the_notebook(oracle, figures, capture):
    scene = nlos_notebook_scene(capture, 1024)
    t4, _ = oracle_render(oracle, scene, 1024, seeds=(0, 1))
    check_nlos_frames(figures, capture, lambda t: t4[:, :, t, 0], ncc_min=0.9, sum_tol=0.08)


@pytest.mark.parametrize("name", ["nlos-z-simple", "nlos-z-room"])
def test_oracle_camera_nlos_frames_match_the_notebook(oracle, figures, name, tmp_path):
    scene = nlos_camera_scene(name, 25000, tmp_path)
    t4, _ = oracle_render(oracle, scene, 25000)
    check_nlos_camera_frames(figures, name, t4[..., 0])


def test_oracle_nlos_pixel_response_matches_the_notebook(oracle, figures):
    scene = nlos_notebook_scene("single", 2048)
    t4, _ = oracle_render(oracle, scene, 2048, seeds=(0, 1, 2, 3))
    check_nlos_pixel_curve(figures, t4[11, 11, :, 0], peak_tol=0.2)


def test_oracle_cornell_box_matches_the_rainbow_figures(oracle, figures):
    scene = cbox_scene(48)
    t4, s4 = oracle_render(oracle, scene, 48)
    t3, s3 = oracle.develop(scene.data().film, t4, s4)
    check_cbox_rainbow(figures, t3)
    check_cbox_steady(figures, s3)


def test_oracle_phasor_film_matches_the_frequency_figures(oracle, figures):
    scene = cbox_scene(128, freq=True)
    film = scene.sensors()[0].film()
    F = len(film.frequencies)
    assert F == 41 and abs(film.frequencies[0][0] - 0.005) < 1e-9 and abs(film.frequencies[-1][0] - 0.015) < 1e-9
    t4, _ = oracle_render(oracle, scene, 128)
    check_cbox_freq(figures, t4[..., :2 * F].reshape(200, 200, F, 2), ncc_min={0: 0.93, 10: 0.88, 20: 0.85, 30: 0.8, 40: 0.75})


def test_figure_tools_read_back_a_known_figure(tmp_path):
    """the digitiser on a figure drawn here from known data: values come back within figure precision"""
    import matplotlib.pyplot as plt
    from PIL import Image
    matplotlib.use("Agg")
    rng = np.random.default_rng(3)
    data = rng.random((64, 64)) ** 3 * 0.37
    plt.figure()
    plt.imshow(data, cmap="hot")
    plt.colorbar()
    plt.axis("off")
    plt.title("t_idx = 1")
    plt.savefig(tmp_path / "f.png", bbox_inches="tight")
    plt.close()
    rgb = np.asarray(Image.open(tmp_path / "f.png").convert("RGB"))
    box = ft.hot_image_box(rgb)
    _, hi = ft.colorbar_range(rgb, 0.05, x_from=box[3] + 3)
    got = ft.invert_cmap(ft.cells(rgb, box, 64, 64), "hot") * hi
    assert abs(hi / data.max() - 1.0) <= 0.02 and np.abs(got - data).max() <= 0.02 * data.max() + 2e-3


# ---------------------------------------------------------------------------------------------------------------------
# GPU: the product, at the notebooks' sample counts
@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["fused", "wavefront"])
@pytest.mark.parametrize("capture", ["single", "confocal"])
def test_product_nlos_frames_match_the_notebook(figures, capture, mode):
    scene = nlos_notebook_scene(capture, 25000)
    scene.integrator().amd_mode = mode
    tr, _ = product_render(scene, 25000)
    assert tr.shape == (64, 64, 300, 1)
    check_nlos_frames(figures, capture, lambda t: tr[:, :, t, 0], ncc_min=0.97, sum_tol=0.06)
    if capture == "single":
        check_nlos_pixel_curve(figures, tr[11, 11, :, 0], peak_tol=0.12)


@pytest.mark.gpu
@pytest.mark.parametrize("name,spp", [("nlos-z-simple", 25000), ("nlos-z-room", 250000)])
def test_product_camera_nlos_frames_match_the_notebook(figures, name, spp, tmp_path):
    scene = nlos_camera_scene(name, spp, tmp_path)
    tr, _ = product_render(scene, spp)
    assert tr.shape == (32, 32, 300, 1)
    check_nlos_camera_frames(figures, name, tr[..., 0])


@pytest.mark.gpu
def test_product_nlos_exhaustive_frames_match_the_notebook(figures):
    """cells 23-25: 32 x 32 scan x 32 x 32 illuminated points; np.array(data)[:, :, laser_x, laser_y, t, 0]"""
    figs, meta = figures
    scene = nlos_notebook_scene("exhaustive", 1500, res=32, exhaustive=True)
    tr, _ = product_render(scene, 1500)
    assert tr.shape == (32, 32, 32, 32, 300, 1)
    for name, m in meta.items():
        if m["kind"] != "hot" or m["capture"] != "exhaustive":
            continue
        box = ft.hot_image_box(figs[name])
        _, hi = ft.colorbar_range(figs[name], m["tick_step"], x_from=box[3] + 3)
        ref = ft.invert_cmap(ft.cells(figs[name], box, 32, 32), "hot") * hi
        lx, ly = m["laser"]
        mine = tr[:, :, lx, ly, m["t"], 0]
        c = ft.ncc(ref, mine)
        assert c >= 0.85, (name, c)
        if lx != ly:
            assert ft.ncc(ref, tr[:, :, ly, lx, m["t"], 0]) <= c - 0.2, name          # the order of the two laser indices
        assert max(ft.ncc(ref, mine[:, ::-1]), ft.ncc(ref, mine[::-1])) <= c - 0.05, name
        assert abs(mine.sum() / ref.sum() - 1.0) <= 0.12, (name, mine.sum(), ref.sum())


@pytest.mark.gpu
def test_product_cornell_box_matches_the_rainbow_figures(figures):
    scene = cbox_scene(1024)
    t3, s3 = product_render(scene, 1024)
    check_cbox_rainbow(figures, t3)
    check_cbox_steady(figures, s3)


@pytest.mark.gpu
def test_product_phasor_film_matches_the_frequency_figures(figures):
    scene = cbox_scene(128, freq=True)
    ph, steady = product_render(scene, 128)
    assert ph.shape == (200, 200, 41, 2) and steady.shape == (200, 200, 1)
    check_cbox_freq(figures, ph, ncc_min={0: 0.93, 10: 0.88, 20: 0.85, 30: 0.8, 40: 0.75})


@pytest.mark.gpu
def test_product_matches_the_readme_images(figures):
    """the PRODUCT against `.images/cornell-box.png` (a render of cbox_diffuse.xml: all three channels of six surfaces) and
    `.images/staircase_steady.png` (config 5's scene as its file describes it; the bench's smooth-material approximation of the
    same scene is measurably NOT that picture)"""
    scene = readme_cornell_scene(356, 1024)
    _, s3 = product_render(scene, 1024)
    ratios = check_readme_cornell_box(figures, s3, ncc_min=0.8, interior_min=0.9, chroma_tol=0.04, level_tol=0.05)
    print("README cornell-box.png / render of cbox_diffuse.xml, per surface (r, g, b):", {k: np.round(v, 3).tolist() for k, v in ratios.items()})
    _, s3d = product_render(readme_cornell_scene(356, 1024, "dict"), 1024)
    with pytest.raises(AssertionError):             # mitransient.cornell_box() is not that picture
        check_readme_cornell_box(figures, s3d, ncc_min=0.8, interior_min=0.9, chroma_tol=0.04, level_tol=0.05)
    scene = readme_staircase_scene(216, 384, 512)
    _, s3 = product_render(scene, 512)
    c_full, ratio = check_readme_staircase(figures, s3, ncc_min=0.94, interior_min=0.98, mean_tol=0.05)
    from mitransient_amd.scenes import staircase
    approx = staircase(width=216, height=384, spp=256, temporal_bins=64)
    _, s3a = product_render(approx, 256)
    with pytest.raises(AssertionError):
        check_readme_staircase(figures, s3a, ncc_min=0.94, interior_min=0.98, mean_tol=0.05)
    fig = figures[0]["readme_staircase_steady"]
    ref = figure_content(fig, bright_margin=True)
    mean_a = (resized(display(s3a), ref.shape[:2]) ** 2.2).reshape(-1, 3).mean(0) / (ref ** 2.2).reshape(-1, 3).mean(0)
    assert mean_a[0] > 1.4 and np.all(np.abs(ratio - 1.0) <= 0.05)


@pytest.mark.gpu
@pytest.mark.parametrize("unwarp", [True, False])
def test_product_matches_the_readme_video_of_the_staircase(figures, unwarp):
    """config 5's TIME AXIS against the reference's own transient video: GIF frame k == time bin k + 20, exactly, in all 31
    sampled frames from the first light to the wavefront reaching the floor; without camera_unwarp (the camera-to-surface leg
    counted) the same check must fail — the video was rendered with the scene file's `camera_unwarp = true`"""
    scene = readme_staircase_scene(216, 384, 256, temporal_bins=400, camera_unwarp=unwarp)
    t3, _ = product_render(scene, 256)
    if unwarp:
        off, scores = check_readme_staircase_video(figures, t3, least_ncc=0.8, min_exact=1.0)
        assert scores[1:12].min() >= 0.9            # (the frames after the very first light: 0.92 ... 0.98)
    else:
        with pytest.raises(AssertionError):
            check_readme_staircase_video(figures, t3, least_ncc=0.8, min_exact=1.0)
