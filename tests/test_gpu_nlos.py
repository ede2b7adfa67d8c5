"""GPU parity of the NLOS tier: k_fused<NLOS> + k_nlos_prepare through the C-ABI against the CPU oracle."""
import numpy as np
import pytest

from conftest import make_nlos, rel_l2
from mitransient_amd import _cabi
from test_nlos import CONFIGS

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _gpu(scene, spp, seed=0):
    import torch
    integ = scene.integrator()
    integ.collect_stats = True
    s, t = integ.render(scene, seed=seed, spp=spp)
    torch.cuda.synchronize()
    return np.array(s), np.array(t)


def _oracle(oracle, scene, spp, seed=0):
    sd = scene.data()
    p = scene.integrator().render_params(scene.sensors()[0].film(), seed, spp)
    t4, s4, c = oracle.render(sd, p, use_bvh=True)
    t3, s3 = oracle.develop(sd.film, t4, s4)
    return s3, t3, c


@pytest.mark.parametrize("mode", [1, 2])            # MTR_MODE_FUSED (k_fused<NLOS>), MTR_MODE_WAVEFRONT (k_wf_nlos_bounce + k_wf_scatter)
@pytest.mark.parametrize("capture,integ", CONFIGS)
@pytest.mark.parametrize("hidden", ["quad", "z"])
def test_nlos_matches_oracle(oracle, capture, integ, hidden, mode):
    scene = make_nlos(sx=8, sy=6, capture=capture, hidden=hidden, **integ)
    scene.integrator().mode = mode
    s_gpu, t_gpu = _gpu(scene, 64)
    s_ref, t_ref, cnt = _oracle(oracle, scene, 64)
    assert t_gpu.shape == (6, 8, 64, 3)
    if np.linalg.norm(t_ref) == 0:
        assert not t_gpu.any()
    else:
        assert rel_l2(t_gpu, t_ref) <= TOL
    assert np.linalg.norm(s_ref) == 0 or rel_l2(s_gpu, s_ref) <= TOL
    got = scene.integrator().last_counters
    for k in ("paths", "rays_closest", "rays_shadow", "splats_issued", "bounces"):
        assert got[k] == cnt[k], k


def test_refocusing_the_laser_between_renders(oracle):
    """mitransient.nlos.focus_emitter_at_relay_wall_pixel moves the laser: the next render must see it
    (mtr_scene_set_nlos re-derives the tables and the scanned points)."""
    import mitransient_amd as mitr
    scene = make_nlos(sx=8, sy=8, capture="single")
    s0, t0 = _gpu(scene, 32)
    sensor = scene.sensors()[0]
    relay = [s for s in scene.shapes() if s.sensor() is sensor][0]
    mitr.nlos.focus_emitter_at_relay_wall_pixel((1, 6), relay, scene.emitters()[0])
    s1, t1 = _gpu(scene, 32)
    s_ref, t_ref, _ = _oracle(oracle, scene, 32)
    assert rel_l2(t1, t_ref) <= TOL
    assert rel_l2(t1, t0) > 0.1


def test_nlos_config4_shape_properties():
    """BASELINE config 4 geometry at reduced size (confocal, T = 4096, bin width 2^-11): energy identity with a window
    that covers everything, and no contribution before 2 x 1.0 of OPL."""
    scene = make_nlos(sx=16, sy=16, capture="confocal", bins=4096, bin_width=2.0 ** -11, start=1.85, hidden="z", max_depth=4)
    s, t = _gpu(scene, 256)
    assert t.shape == (16, 16, 4096, 3)
    first_bin = int(np.nonzero(t.sum(axis=(0, 1, 3)))[0][0])
    assert 1.85 + first_bin * 2.0 ** -11 >= 1.99
    # the wavefront organisation of this tier renders the same lanes (T = 4096: the 48 KB-row instantiation vs records + scatter)
    scene.integrator().mode = 2
    s2, t2 = _gpu(scene, 256)
    assert rel_l2(t2, t) <= 1e-6 and np.array_equal(t2 != 0, t != 0)


# ---- Exhaustive captures + the 6-D exhaustive_scan film -----------------------------------------------------
from test_nlos import EXH, exhaustive_scene  # noqa: E402


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("cfg", EXH)
def test_exhaustive_matches_oracle(oracle, cfg, mode):
    scene = exhaustive_scene(**cfg)
    scene.integrator().mode = mode
    film = scene.sensors()[0].film()
    s_gpu, t_gpu = _gpu(scene, 48)
    sd = scene.data()
    f = sd.film
    p = scene.integrator().render_params(film, 0, 48)
    t6, s4, cnt = oracle.render(sd, p, use_bvh=True)
    t_ref, _ = oracle.develop(sd.film, t6, None)
    assert t_gpu.shape == (f.height, f.width, f.laser_scan_height, f.laser_scan_width, f.temporal_bins, 3)
    assert rel_l2(t_gpu, t_ref) <= TOL
    # transient_hdr_film.py:213-214: the "steady" image of an exhaustive film is mean(transient, axis=-1)
    assert s_gpu.shape == t_gpu.shape[:-1] and np.allclose(s_gpu, t_gpu.mean(axis=-1), rtol=1e-6, atol=1e-12)
    _, raw = film.develop(raw=True)
    raw = np.array(raw)
    assert raw.shape == t6.shape and np.array_equal(raw[..., :3] != 0, t6[..., :3] != 0) and not raw[..., 3].any()
    got = scene.integrator().last_counters
    for k in ("paths", "rays_closest", "rays_shadow", "splats_issued", "bounces"):
        assert got[k] == cnt[k], k


@pytest.mark.parametrize("variant,sort", [(0, False), (1, True)])
def test_exhaustive_film_add_transient_data(oracle, variant, sort):
    """film.add_transient_data(pos, distance, wavelengths, spec, ray_weight, active, laser_x, laser_y) on an
    exhaustive_scan film: the reference's flat index ((((y*W+x)*Lw + laser_x)*Lh + laser_y)*T + t)*C."""
    import torch
    import mitransient_amd as mitr
    from mitransient_amd.scene import Properties
    W, H, Lw, Lh, T = 5, 4, 3, 2, 32
    film = mitr.TransientHDRFilm(Properties("transient_hdr_film", {
        "width": W, "height": H, "temporal_bins": T, "bin_width_opl": 0.1, "start_opl": 1.0, "rfilter": {"type": "box"},
        "exhaustive_scan": True, "laser_scan_width": Lw, "laser_scan_height": Lh}))
    film.prepare([])
    rng = np.random.default_rng(7)
    n = 20000
    pos = rng.uniform([-0.5, -0.5], [W + 0.5, H + 0.5], size=(n, 2)).astype(np.float32)
    if sort:
        order = np.lexsort((pos[:, 0].astype(int), pos[:, 1].astype(int)))
        pos = pos[order]
    dist = rng.uniform(0.8, 4.4, n).astype(np.float32)
    spec = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    lx = rng.integers(0, Lw + 1, n)              # Lw / Lh themselves are out of range -> dropped
    ly = rng.integers(0, Lh + 1, n)
    film.add_transient_data(pos, dist, None, spec, 1.0, None, lx, ly, variant=variant)
    torch.cuda.synchronize()
    _, raw = film.develop(raw=True)
    raw = np.array(raw)
    assert raw.shape == (H, W, Lh, Lw, T, 4)
    px, py = np.floor(pos[:, 0]).astype(np.int64), np.floor(pos[:, 1]).astype(np.int64)
    ok = (px >= 0) & (px < W) & (py >= 0) & (py < H) & (lx < Lw) & (ly < Lh)
    ref = np.zeros(raw.shape, np.float32)
    oracle.splat_add(film.desc(), (py * W + px)[ok], dist[ok], spec[ok, 0], spec[ok, 1], spec[ok, 2], ref, lx[ok], ly[ok])
    assert np.count_nonzero(ref) > 1000
    assert np.allclose(raw, ref, rtol=1e-5, atol=1e-6)
    # independent statement of the index: memory order is [y][x][laser_x][laser_y][t]
    flat = raw.reshape(H, W, Lw, Lh, T, 4)
    i = int(np.argmax(ok & (dist > 1.0) & (dist < 4.2)))
    b = int(np.floor((np.float32(dist[i]) - np.float32(1.0)) / np.float32(0.1)))
    assert flat[py[i], px[i], lx[i], ly[i], b, 0] > 0


from test_nlos import CAMERA_NLOS  # noqa: E402
from conftest import make_nlos_camera  # noqa: E402


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("cfg", CAMERA_NLOS)
def test_camera_nlos_matches_oracle(oracle, cfg, mode):
    """transient_nlos_path with a perspective camera instead of a nlos_capture_meter (nlos-z-simple.xml)"""
    scene = make_nlos_camera(res=20, **cfg)
    scene.integrator().mode = mode
    s_gpu, t_gpu = _gpu(scene, 48)
    s_ref, t_ref, cnt = _oracle(oracle, scene, 48)
    assert t_gpu.shape == (20, 20, 100, 3)
    assert rel_l2(t_gpu, t_ref) <= TOL
    assert np.linalg.norm(s_ref) == 0 or rel_l2(s_gpu, s_ref) <= TOL
    got = scene.integrator().last_counters
    for k in ("paths", "rays_closest", "rays_shadow", "splats_issued", "bounces"):
        assert got[k] == cnt[k], k


@pytest.mark.parametrize("capture", ["single", "confocal", "exhaustive"])
def test_notebook_flow(tmp_path, oracle, capture):
    """examples/transient-nlos/1-simple-nlos-scenes.ipynb end to end: llvm_ad_mono, stand-alone plugins assembled into a
    scene, mi.render(scene) with the sensor's own sample count, (H,W,T,1) / (H,W,Lh,Lw,T,1) outputs"""
    import mitransient_amd as mitr
    import mitransient_amd.mi as mi
    from test_nlos import notebook_scene
    mi.set_variant("llvm_ad_mono")
    try:
        extra = {"exhaustive_scan": True, "laser_scan_width": 8, "laser_scan_height": 8} if capture == "exhaustive" else None
        scene, relay_wall, emitter, film, sensor = notebook_scene(tmp_path, capture, extra, spp=24)
        if capture == "single":
            mitr.nlos.focus_emitter_at_relay_wall_pixel(mi.Point2f(4, 4), relay_wall, emitter)
        data_steady, data_transient = mi.render(scene)
        assert type(data_transient).__name__ == "TensorXf"
        t = np.array(data_transient)
        assert t.shape == ((8, 8, 8, 8, 300, 1) if capture == "exhaustive" else (8, 8, 300, 1))
        sd = scene.data()
        p = scene.integrator().render_params(film, 0, 24)
        t4, s4, cnt = oracle.render(sd, p, use_bvh=True)
        t3, _ = oracle.develop(sd.film, t4, None)
        assert np.linalg.norm(t3) > 0 and rel_l2(t[..., 0], t3[..., 0]) <= TOL
    finally:
        mi.set_variant("llvm_ad_rgb")


def test_is_confocal_capture_meter_gpu(oracle):
    """1 x 1 film + original_film_*: every sensor ray goes to the laser's focus point (nloscapturemeter.py:111-119, :142)"""
    scene = make_nlos(sx=1, sy=1, capture="single", spp=2048, focus=(2.5, 6.5), bins=128, bin_width=0.02, hidden="z",
                      sensor_extra={"original_film_width": 8, "original_film_height": 8}, max_depth=5)
    s_gpu, t_gpu = _gpu(scene, 2048)
    s_ref, t_ref, cnt = _oracle(oracle, scene, 2048)
    assert t_gpu.shape == (1, 1, 128, 3) and np.count_nonzero(t_ref) > 20
    assert rel_l2(t_gpu, t_ref) <= TOL
    got = scene.integrator().last_counters
    for k in ("paths", "rays_closest", "rays_shadow", "splats_issued", "bounces"):
        assert got[k] == cnt[k], k


# ---- the extended shading code in the NLOS tier (VERDICT r2 task 8): vertex normals, GGX lobes, bitmaps on hidden geometry -------
@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("capture,integ", [("confocal", {}), ("single", {"nlos_hidden_geometry_sampling": False}), ("exhaustive", {})])
@pytest.mark.parametrize("bsdf", ["diffuse", "roughconductor", "roughplastic", "twosided-roughplastic"])
def test_hidden_mesh_with_vertex_normals_and_rough_lobes_gpu(oracle, tmp_path, bsdf, capture, integ, mode):
    from test_nlos import ROUGH_HIDDEN, _hidden_sphere
    kw = dict(sx=6, sy=5, capture=capture, hidden=_hidden_sphere(tmp_path, ROUGH_HIDDEN[bsdf]), **integ)
    if capture == "exhaustive":
        kw.update(film={"exhaustive_scan": True, "laser_scan_width": 6, "laser_scan_height": 5}, bins=48, bin_width=0.05, start=1.8,
                  force_equal_illumination_scanning=True)
    scene = make_nlos(**kw)
    assert scene.data().tri_normals is not None
    scene.integrator().mode = mode
    s_gpu, t_gpu = _gpu(scene, 64)
    if capture == "exhaustive":
        sd = scene.data()
        t6, s4, cnt = oracle.render(sd, scene.integrator().render_params(scene.sensors()[0].film(), 0, 64), use_bvh=True)
        t_ref, s_ref = oracle.develop(sd.film, t6, None)[0], None
    else:
        s_ref, t_ref, cnt = _oracle(oracle, scene, 64)
    assert np.count_nonzero(t_ref) > 20 and rel_l2(t_gpu, t_ref) <= TOL
    # normals without lobes: the extended kernel built without the lobe code (k_fused<..., kTrNoLobes>) — the same numbers
    assert bool(scene.gpu_traits() & _cabi.MTR_TRAIT_NO_LOBES) == (bsdf == "diffuse")
    if capture != "exhaustive":              # (an exhaustive film's steady image is the mean over time: test_exhaustive_matches_oracle)
        assert np.linalg.norm(s_ref) == 0 or rel_l2(s_gpu, s_ref) <= TOL
    got = scene.integrator().last_counters
    for k in ("paths", "rays_closest", "rays_shadow", "splats_issued", "bounces"):
        assert got[k] == cnt[k], k


@pytest.mark.parametrize("mode", [1, 2])
def test_textured_hidden_geometry_gpu(oracle, tmp_path, mode):
    from test_textures import make_texture
    make_texture(str(tmp_path / "tex.png"))
    tex = {"type": "diffuse", "reflectance": {"type": "bitmap", "filename": str(tmp_path / "tex.png")}}
    scene = make_nlos(sx=8, sy=8, capture="confocal", hidden="quad", hidden_bsdf=tex)
    assert len(scene.data().textures) == 1
    scene.integrator().mode = mode
    s_gpu, t_gpu = _gpu(scene, 64)
    s_ref, t_ref, cnt = _oracle(oracle, scene, 64)
    assert np.count_nonzero(t_ref) > 50 and rel_l2(t_gpu, t_ref) <= TOL
    assert scene.gpu_traits() & _cabi.MTR_TRAIT_NO_LOBES
    got = scene.integrator().last_counters
    for k in ("paths", "rays_closest", "rays_shadow", "splats_issued", "bounces"):
        assert got[k] == cnt[k], k


def test_extended_nlos_deterministic_rows_take_the_wavefront_pipeline(tmp_path):
    """deterministic rows + the extended shading: AUTO resolves to the wavefront organisation (the fused kernel's fixed-point
    rows are built without the extended code), two renders are bit-identical; asking for the fused kernel is refused"""
    from test_nlos import ROUGH_HIDDEN, _hidden_sphere
    from mitransient_amd._cabi import MitransientAMDError
    def build(**kw):
        return make_nlos(sx=4, sy=4, capture="confocal", hidden=_hidden_sphere(tmp_path, ROUGH_HIDDEN["roughplastic"]), **kw)
    a = build(amd_deterministic=True); b = build(amd_deterministic=True)
    ta, tb = _gpu(a, 32)[1], _gpu(b, 32)[1]
    assert np.array_equal(ta, tb) and np.count_nonzero(ta) > 20
    assert a.integrator().resolved_mode(a, a.sensors()[0], 32) == "wavefront"
    c = build(amd_deterministic=True, amd_mode="fused")
    with pytest.raises(MitransientAMDError, match="wavefront"):
        _gpu(c, 8)


# ---- scene trait kTrGrey: a grey NLOS scene keeps ONE plane per row in k_fused<NLOS> (config 4: three row slots instead of one) --------
COLOURED = {"type": "diffuse", "reflectance": {"type": "rgb", "value": [0.9, 0.5, 0.2]}}


@pytest.mark.parametrize("hidden", ["quad", "z", "sphere"])
def test_grey_scene_one_plane_per_row(oracle, tmp_path, hidden):
    """the premise — every contribution of a grey scene has r == g == b to the bit — holds in the organisation that keeps three channels
    (wavefront) and in the oracle; the fused kernel's one-plane rows give the oracle's film and counters"""
    from test_nlos import ROUGH_HIDDEN, _hidden_sphere
    h = _hidden_sphere(tmp_path, ROUGH_HIDDEN["diffuse"]) if hidden == "sphere" else hidden        # (vertex normals: the extended kernel without lobes)
    films = {}
    for mode in (1, 2):
        scene = make_nlos(sx=8, sy=6, capture="confocal", hidden=h, bins=96, bin_width=0.03, start=1.8)
        assert scene.gpu_traits() & _cabi.MTR_TRAIT_GREY
        scene.integrator().mode = mode
        s_gpu, t_gpu = _gpu(scene, 128)
        assert np.array_equal(t_gpu[..., 0], t_gpu[..., 1]) and np.array_equal(t_gpu[..., 0], t_gpu[..., 2]) and np.count_nonzero(t_gpu) > 50
        films[mode] = t_gpu
        s_ref, t_ref, cnt = _oracle(oracle, scene, 128)
        assert np.array_equal(t_ref[..., 0], t_ref[..., 1]) and np.array_equal(t_ref[..., 0], t_ref[..., 2])
        assert rel_l2(t_gpu, t_ref) <= TOL and rel_l2(s_gpu, s_ref) <= TOL
        got = scene.integrator().last_counters
        for k in ("paths", "rays_closest", "rays_shadow", "splats_issued", "bounces"):
            assert got[k] == cnt[k], k
    assert rel_l2(films[1], films[2]) <= 1e-6


@pytest.mark.parametrize("what", ["hidden", "laser", "wall-texture"])
def test_coloured_scene_keeps_three_planes(oracle, tmp_path, what):
    """one colour anywhere — a reflectance, the laser, a bitmap — and the trait is off: k_fused<NLOS> with rgb rows, the oracle's film"""
    kw = dict(sx=8, sy=6, capture="confocal", hidden="quad", bins=96, bin_width=0.03, start=1.8)
    if what == "hidden":
        scene = make_nlos(hidden_bsdf=COLOURED, **kw)
    elif what == "laser":
        scene = make_nlos(laser_rgb=(1.0, 0.6, 0.3), **kw)
    else:
        from test_textures import make_texture
        make_texture(str(tmp_path / "tex.png"))
        scene = make_nlos(hidden_bsdf={"type": "diffuse", "reflectance": {"type": "bitmap", "filename": str(tmp_path / "tex.png")}}, **kw)
    scene.integrator().mode = 1
    s_gpu, t_gpu = _gpu(scene, 128)
    assert not (scene.gpu_traits() & _cabi.MTR_TRAIT_GREY)
    s_ref, t_ref, cnt = _oracle(oracle, scene, 128)
    assert not np.array_equal(t_ref[..., 0], t_ref[..., 2]) and np.count_nonzero(t_ref) > 50
    assert rel_l2(t_gpu, t_ref) <= TOL and rel_l2(s_gpu, s_ref) <= TOL
    got = scene.integrator().last_counters
    for k in ("paths", "rays_closest", "rays_shadow", "splats_issued", "bounces"):
        assert got[k] == cnt[k], k
