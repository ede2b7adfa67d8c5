"""phasor_hdr_film (mitransient/films/phasor_hdr_film.py, render/phasor_image_block.py) — CPU tests:
the oracle's restatement against independent pins, product arithmetic (host harness) == oracle bit for bit."""
import numpy as np
import pytest

from conftest import hh_render, rel_l2


@pytest.fixture
def mono():
    import mitransient_amd.mi as mi
    mi.set_variant("llvm_ad_mono")
    yield mi
    mi.set_variant("llvm_ad_rgb")


def phasor_cornell(mi, res=12, **film):
    import mitransient_amd as mitr
    d = mitr.cornell_box()
    fd = {"type": "phasor_hdr_film", "width": res, "height": res, "wl_mean": 2.0, "wl_sigma": 1.0, "temporal_bins": 400,
          "bin_width_opl": 0.05, "start_opl": 3.0, "rfilter": {"type": "box"}}
    fd.update(film)
    d["sensor"]["film"] = fd
    return mi.load_dict(d)


def test_phasor_term_against_float64(oracle):
    rng = np.random.default_rng(3)
    f = rng.uniform(0.001, 2.0, 4000).astype(np.float32)
    opl = rng.uniform(-50.0, 400.0, 4000).astype(np.float32)
    got = np.array([oracle.phasor_term(a, b) for a, b in zip(f, opl)])
    # the reference's f32 phase (phasor_image_block.py:49-52), then float64 cos / sin of it
    x = (np.float32(-2 * np.pi) * f) * opl
    y = np.float32(2 * np.pi)
    phase = (x - y * np.floor(x / y)).astype(np.float64)
    assert np.abs(got[:, 0] - np.cos(phase)).max() < 3e-7 and np.abs(got[:, 1] - np.sin(phase)).max() < 3e-7
    assert oracle.phasor_term(0.25, 0.0) == (1.0, 0.0)


def test_frequencies_and_plugin_surface(mono):
    import mitransient_amd as mitr
    from mitransient_amd.scene import Properties
    f = mitr.PhasorHDRFilm(Properties("phasor_hdr_film", {"width": 8, "height": 8, "wl_mean": 100, "wl_sigma": 100,
                                                          "temporal_bins": 4000, "bin_width_opl": 1.0, "rfilter": {"type": "box"}}))
    # examples/transient/cornell-box/cbox_diffuse_freq.xml: indices 20..60 of fftfreq(4000, 1)
    assert len(f.frequencies) == 41 and f.frequencies[0] == np.float32(20 / 4000) and f.frequencies[-1] == np.float32(60 / 4000)
    assert f.raw_shape() == (8, 8, 83)
    d = mitr.PhasorHDRFilm(Properties("phasor_hdr_film", {}))
    assert (d.wl_mean, d.wl_sigma, d.temporal_bins, d.bin_width_opl, d.start_opl) == (100.0, 1000.0, 4096, 0.003, 0.0)
    with pytest.raises(ValueError):
        mitr.PhasorHDRFilm(Properties("phasor_hdr_film", {"width": 8, "height": 8, "crop_width": 4}))
    scene = phasor_cornell(mono)
    assert isinstance(scene.sensors()[0].film(), mitr.PhasorHDRFilm)
    sd = scene.data()
    m = [sd.materials[i] for i in range(sd.n_materials)]
    assert all(x.a[0] == x.a[1] == x.a[2] for x in m)                      # colours -> luminance in the mono variants
    lum = np.float32(np.float32(0.570068) * np.float32(0.212671) + np.float32(0.0430135) * np.float32(0.715160)) \
        + np.float32(0.0443706) * np.float32(0.072169)
    assert any(abs(x.a[0] - lum) < 1e-7 for x in m)                         # the red wall


def test_phasor_render_harness_equals_oracle_and_matches_dft_of_the_histogram(mono, oracle, host_harness):
    scene = phasor_cornell(mono)
    sd = scene.data()
    film = scene.sensors()[0].film()
    F = len(film.frequencies)
    assert F > 20
    p = scene.integrator().render_params(film, 0, 32)
    t, s4, cnt = oracle.render(sd, p, n_threads=1)
    assert t.shape == (12, 12, 2 * F + 1) and not t[..., -1].any()
    ht, hs, hc = hh_render(host_harness, sd, p)
    assert np.array_equal(t, ht) and np.array_equal(s4, hs)
    for k in ("paths", "rays_closest", "rays_shadow", "splats_issued", "bounces"):
        assert hc[k] == cnt[k], k
    ph, _ = oracle.develop(sd.film, t, None)
    assert ph.shape == (12, 12, F, 2)
    # independent pin: the same lanes splatted into a fine time histogram, then a direct Fourier sum per frequency
    from mitransient_amd import _cabi
    import copy
    sd2 = copy.copy(sd)
    fd = _cabi.mtr_film_desc()
    fd.width = fd.crop_width = 12
    fd.height = fd.crop_height = 12
    fd.temporal_bins, fd.start_opl, fd.bin_width_opl = 8000, np.float32(3.0), np.float32(0.0025)
    sd2.film = fd
    t4, _, cnt2 = oracle.render(sd2, p, n_threads=1)
    assert cnt2["paths"] == cnt["paths"] and cnt2["bounces"] == cnt["bounces"]
    hist = t4[..., 0].astype(np.float64)                                    # (H, W, T)
    assert hist.sum() > 0 and abs(hist.sum() / float(t4[..., 1].sum()) - 1) < 1e-6      # three equal channels
    centres = (np.arange(8000) + 0.5) * 0.0025
    fr = np.asarray(film.frequencies, np.float64)
    ker = np.exp(-2j * np.pi * fr[None, :] * centres[:, None])              # exp(i * (-2 pi f opl))
    ref = hist @ ker
    got = ph[..., 0].astype(np.float64) + 1j * ph[..., 1].astype(np.float64)
    # all of this scene's optical path lengths fall inside the 3 .. 23 window of the histogram; binning error ~ (pi f w)^2
    assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < 2e-3
