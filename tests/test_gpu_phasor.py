"""phasor_hdr_film on the GPU: wavefront records -> k_wf_phasor_scatter, through the C-ABI, against the CPU oracle."""
import numpy as np
import pytest

from conftest import rel_l2
from test_phasor import phasor_cornell, mono  # noqa: F401

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _render(scene, spp, seed=0):
    import torch
    integ = scene.integrator()
    integ.collect_stats = True
    steady, phasors = integ.render(scene, seed=seed, spp=spp)
    torch.cuda.synchronize()
    return np.array(steady), np.array(phasors)


def _oracle(oracle, scene, spp, seed=0):
    sd = scene.data()
    p = scene.integrator().render_params(scene.sensors()[0].film(), seed, spp)
    t, s4, cnt = oracle.render(sd, p, use_bvh=True)
    ph, s3 = oracle.develop(sd.film, t, s4)
    return s3, ph, t, cnt


@pytest.mark.parametrize("mode", [0, 1])        # MTR_MODE_AUTO (wavefront: records + k_wf_phasor_scatter), MTR_MODE_FUSED (LDS (Re, Im) rows)
@pytest.mark.parametrize("res,spp,film", [(16, 64, {}), (9, 300, {"wl_mean": 1.0, "wl_sigma": 0.2}),
                                          (24, 2, {})])           # 2 spp: 8-record lists overflow -> the atomic fallback
def test_phasor_render_matches_oracle(mono, oracle, res, spp, film, mode):
    scene = phasor_cornell(mono, res=res, **film)
    scene.integrator().mode = mode
    F = len(scene.sensors()[0].film().frequencies)
    s_gpu, p_gpu = _render(scene, spp, seed=2)
    s_ref, p_ref, raw_ref, cnt = _oracle(oracle, scene, spp, seed=2)
    assert p_gpu.shape == (res, res, F, 2) and s_gpu.shape == (res, res, 1)
    assert rel_l2(p_gpu, p_ref) <= TOL and rel_l2(s_gpu[..., 0], s_ref[..., 0]) <= TOL
    got = scene.integrator().last_counters
    for k in ("paths", "rays_closest", "rays_shadow", "splats_issued", "bounces"):
        assert got[k] == cnt[k], k
    if spp == 2 and mode == 0:
        assert got["splats_overflow"] > 0
    _, raw = scene.sensors()[0].film().develop(raw=True)
    raw = np.array(raw)
    assert raw.shape == (res, res, 2 * F + 1) and not raw[..., -1].any() and rel_l2(raw, raw_ref) <= TOL
    # the zero frequency would be the steady image; here: |phasor| never exceeds the total energy of the pixel
    assert np.all(np.hypot(p_gpu[..., 0], p_gpu[..., 1]).max(axis=2) <= s_gpu[..., 0] * (1 + 1e-3) + 1e-6)


def test_phasor_add_transient_data_and_errors(mono, oracle):
    import torch
    import mitransient_amd as mitr
    from mitransient_amd.scene import Properties
    film = mitr.PhasorHDRFilm(Properties("phasor_hdr_film", {"width": 6, "height": 5, "wl_mean": 4.0, "wl_sigma": 2.0,
                                                              "temporal_bins": 256, "bin_width_opl": 0.1, "start_opl": 1.5,
                                                              "rfilter": {"type": "box"}}))
    film.prepare([])
    rng = np.random.default_rng(11)
    n = 5000
    pos = rng.uniform([-0.5, -0.5], [6.5, 5.5], size=(n, 2)).astype(np.float32)
    dist = rng.uniform(0.0, 30.0, n).astype(np.float32)
    dist[:5] = np.inf
    spec = rng.uniform(0, 1, n).astype(np.float32)
    film.add_transient_data(pos, dist, None, spec, 1.0, None)
    torch.cuda.synchronize()
    _, raw = film.develop(raw=True)
    raw = np.array(raw)
    px, py = np.floor(pos[:, 0]).astype(np.int64), np.floor(pos[:, 1]).astype(np.int64)
    ok = (px >= 0) & (px < 6) & (py >= 0) & (py < 5)
    ref = np.zeros_like(raw)
    oracle.splat_add(film.desc(), (py * 6 + px)[ok], dist[ok], spec[ok], spec[ok], spec[ok], ref)
    assert np.count_nonzero(ref) > 500 and np.allclose(raw, ref, rtol=1e-4, atol=1e-4)
    # errors: rgb variant
    import mitransient_amd.mi as mi
    mi.set_variant("llvm_ad_rgb")
    with pytest.raises(RuntimeError, match="monochromatic"):
        sc = phasor_cornell(mi)
        sc.integrator().render(sc, spp=1)
    mi.set_variant("llvm_ad_mono")


def test_mono_transient_film(mono, oracle):
    """llvm_ad_mono with transient_hdr_film (the variant of the reference's NLOS notebooks): one colour channel,
    raw layout "LW"; equals the luminance-coloured RGB run"""
    import mitransient_amd as mitr
    d = mitr.cornell_box()
    d["sensor"]["film"].update(width=16, height=16, temporal_bins=64, bin_width_opl=6.0 / 64)
    sc = mono.load_dict(d)
    steady, transient = sc.integrator().render(sc, spp=8)
    steady, transient = np.array(steady), np.array(transient)
    assert transient.shape == (16, 16, 64, 1) and steady.shape == (16, 16, 1)
    _, raw = sc.sensors()[0].film().develop(raw=True)
    raw = np.array(raw)
    assert raw.shape == (16, 16, 64, 2) and not raw[..., 1].any() and sc.sensors()[0].film().channels == ["L", "W"]
    sd = sc.data()
    p = sc.integrator().render_params(sc.sensors()[0].film(), 0, 8)
    t4, s4, _ = oracle.render(sd, p, use_bvh=True)
    t3, s3 = oracle.develop(sd.film, t4, s4)
    assert np.array_equal(t3[..., 0], t3[..., 1]) and np.array_equal(t3[..., 0], t3[..., 2])      # channels never mix
    assert rel_l2(transient[..., 0], t3[..., 0]) <= TOL and rel_l2(steady[..., 0], s3[..., 0]) <= TOL


def test_variant1_scratch_survives_a_phasor_splat(mono, oracle):
    """a phasor splat that grows the ctx's frequency buffer once also freed the variant-1 run table without
    resetting it — the next sorted-by-pixel splat then wrote through a dangling pointer.  Sequence: variant 1 on a
    transient film, a phasor splat with more frequencies than ever before, variant 1 again; both must match the oracle."""
    import torch
    import mitransient_amd as mitr
    from mitransient_amd.scene import Properties
    rng = np.random.default_rng(5)

    def sorted_splat():
        film = mitr.TransientHDRFilm(Properties("transient_hdr_film", {"width": 16, "height": 8, "temporal_bins": 128,
                                                                        "bin_width_opl": 0.05, "start_opl": 1.0,
                                                                        "rfilter": {"type": "box"}}))
        film.prepare([])
        n = 30000
        pix = np.sort(rng.integers(0, 16 * 8, n)).astype(np.int64)
        pos = np.stack([(pix % 16) + 0.5, (pix // 16) + 0.5], axis=1).astype(np.float32)
        dist = rng.uniform(0.9, 7.6, n).astype(np.float32)
        spec = rng.uniform(0, 1, (n, 3)).astype(np.float32)
        film.add_transient_data(pos, dist, None, spec, 1.0, None, variant=1)
        torch.cuda.synchronize()
        got = np.array(film.develop()[1])                           # monochromatic variant: (H, W, T, 1) = channel 0
        ref = np.zeros((8, 16, 128, 4), np.float32)                 # the accumulator is RGBW in every variant
        oracle.splat_add(film.desc(), pix, dist, spec[:, 0], spec[:, 1], spec[:, 2], ref)
        assert got.shape == (8, 16, 128, 1) and np.count_nonzero(ref) > 1000 and rel_l2(got[..., 0], ref[..., 0]) <= TOL

    sorted_splat()
    for sigma in (1.0, 3.0):                     # the second film has more frequencies: the ctx buffer grows again
        ph = mitr.PhasorHDRFilm(Properties("phasor_hdr_film", {"width": 4, "height": 4, "wl_mean": 4.0, "wl_sigma": sigma,
                                                                "temporal_bins": 512, "bin_width_opl": 0.1, "start_opl": 0.0,
                                                                "rfilter": {"type": "box"}}))
        ph.prepare([])
        ph.add_transient_data(np.full((64, 2), 1.5, np.float32), rng.uniform(0, 9, 64).astype(np.float32), None,
                              np.ones(64, np.float32), 1.0, None)
        torch.cuda.synchronize()
        sorted_splat()
