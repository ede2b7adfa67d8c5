"""GPU parity tests: the HIP path (through the C-ABI) against the CPU oracle, same seeds.

Bar (BASELINE.json north_star): relative L2 <= 1e-5 on the (H,W,T,3) tensor; integer
outputs (counters, (pixel,bin) sets) exact.
"""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import make_cornell, rel_l2

pytestmark = pytest.mark.gpu

TOL = 1e-5   # relative L2, stated by BASELINE.json north_star
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def gpu_render(scene, spp, seed=0, raw=False, **kw):
    import torch
    import mitransient_amd.mi as mi
    integ = scene.integrator()
    integ.collect_stats = True
    steady, transient = integ.render(scene, seed=seed, spp=spp, **kw)
    torch.cuda.synchronize()
    if raw:
        s_raw, t_raw = scene.sensors()[0].film().develop(raw=True)
        return np.array(steady), np.array(transient), np.array(s_raw), np.array(t_raw)
    return np.array(steady), np.array(transient)


def oracle_render(oracle, scene, spp, seed=0, spp_range=None, pixel_range=None):
    integ = scene.integrator()
    film = scene.sensors()[0].film()
    sd = scene.data()
    s0, s1 = (0, spp) if spp_range is None else spp_range
    p0, p1 = (0, None) if pixel_range is None else pixel_range
    params = integ.render_params(film, seed, spp, s0, s1, p0, p1)
    t4, s4, cnt = oracle.render(sd, params, use_bvh=True)
    t3, s3 = oracle.develop(sd.film, t4, s4)
    return s3, t3, s4, t4, cnt


def test_library_is_native_and_loaded():
    from mitransient_amd import _cabi
    lib = _cabi.load_library()
    assert lib.mtr_abi_version() == _cabi.MTR_ABI_VERSION
    with open("/proc/self/maps") as fh:
        assert "libmitransient_amd.so" in fh.read()


MODES = ["fused", "wavefront"]


@pytest.mark.parametrize("mode", MODES)
def test_config1_matches_oracle(oracle, mode):
    """BASELINE config 1: Cornell 64x64, 64 bins, 16 spp — both kernel organisations."""
    scene = make_cornell(amd_mode=mode)
    s_gpu, t_gpu, s_raw, t_raw = gpu_render(scene, 16, raw=True)
    s_ref, t_ref, s4, t4, cnt = oracle_render(oracle, scene, 16)
    assert t_gpu.shape == (64, 64, 64, 3) and s_gpu.shape == (64, 64, 3)
    assert t_raw.shape == (64, 64, 64, 4) and np.all(t_raw[..., 3] == 0)       # "W" channel stays 0
    assert rel_l2(t_gpu, t_ref) <= TOL
    assert rel_l2(s_gpu, s_ref) <= TOL
    # the set of touched (pixel, bin, channel) cells is identical
    assert np.array_equal(t_raw[..., :3] != 0, t4[..., :3] != 0)
    got = scene.integrator().last_counters
    for k in ("paths", "rays_closest", "rays_shadow", "splats_issued", "bounces"):
        assert got[k] == cnt[k], k


@pytest.mark.parametrize("seed", [1, 12345])
def test_seeds(oracle, seed):
    scene = make_cornell(width=32, height=32, bins=128)
    s_gpu, t_gpu = gpu_render(scene, 8, seed=seed)
    s_ref, t_ref, *_ = oracle_render(oracle, scene, 8, seed=seed)
    assert rel_l2(t_gpu, t_ref) <= TOL and rel_l2(s_gpu, s_ref) <= TOL


@pytest.mark.parametrize("mode", MODES)
def test_splat_log_matches_oracle(oracle, mode):
    """splat-for-splat equality: same (lane, depth, kind, pixel, bin) multiset, same values bit for bit."""
    import torch
    from mitransient_amd.runtime import get_context
    scene = make_cornell(width=16, height=16, bins=64, amd_mode=mode)
    integ = scene.integrator()
    film = scene.sensors()[0].film()
    passes = integ.prepare(scene, scene.sensors()[0], 0, 4, [])
    ctx = get_context()
    h = scene.gpu_handle(ctx, 0)
    cap = 1 << 16
    log = torch.zeros((cap, 8), dtype=torch.int32, device="cuda")
    n = torch.zeros(1, dtype=torch.int64, device="cuda")
    ctx.check(ctx.lib.mtr_debug_set_splat_log(h, C.c_void_p(log.data_ptr()), cap, C.c_void_p(n.data_ptr())))
    integ.accumulate(scene, scene.sensors()[0], passes, 4)
    torch.cuda.synchronize()
    ctx.check(ctx.lib.mtr_debug_set_splat_log(h, None, 0, None))
    n_gpu = int(n.item())
    rec = log[:n_gpu].cpu().numpy().view(np.uint32)
    sd = scene.data()
    params = integ.render_params(film, 0, 4)
    _, _, cnt, olog = oracle.render(sd, params, use_bvh=False, log_capacity=cap)
    assert n_gpu == len(olog) == cnt["splats_issued"]
    g = np.zeros(n_gpu, dtype=olog.dtype)
    g["lane"], g["depth_kind"], g["pixel"], g["bin"] = rec[:, 0], rec[:, 1], rec[:, 2], rec[:, 3]
    g["r"], g["g"], g["b"], g["opl"] = (rec[:, 4].view(np.float32), rec[:, 5].view(np.float32),
                                       rec[:, 6].view(np.float32), rec[:, 7].view(np.float32))
    order = ["lane", "depth_kind"]
    g.sort(order=order)
    olog.sort(order=order)
    for k in olog.dtype.names:
        assert np.array_equal(g[k].view(np.uint32), olog[k].view(np.uint32)), k


@pytest.mark.parametrize("mode", MODES)
def test_camera_unwarp_and_discard_direct(oracle, mode):
    scene = make_cornell(width=32, height=32, bins=64, start=0.0, window=8.0, camera_unwarp=True,
                         discard_direct_light=True, amd_mode=mode)
    s_gpu, t_gpu = gpu_render(scene, 8)
    s_ref, t_ref, *_ = oracle_render(oracle, scene, 8)
    assert rel_l2(t_gpu, t_ref) <= TOL and rel_l2(s_gpu, s_ref) <= TOL


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("max_depth,rr_depth", [(0, 5), (1, 5), (2, 5), (3, 1), (12, 2), (-1, 3)])
def test_depths(oracle, max_depth, rr_depth, mode):
    scene = make_cornell(width=24, height=24, bins=64, max_depth=max_depth, rr_depth=rr_depth, amd_mode=mode)
    s_gpu, t_gpu = gpu_render(scene, 8)
    s_ref, t_ref, *_ = oracle_render(oracle, scene, 8)
    if np.linalg.norm(t_ref) == 0:
        assert np.all(t_gpu == 0)
    else:
        assert rel_l2(t_gpu, t_ref) <= TOL
    assert rel_l2(s_gpu, s_ref) <= TOL or np.linalg.norm(s_ref) == 0


@pytest.mark.parametrize("mode", MODES)
def test_ragged_sizes_and_crop(oracle, mode):
    """non-square film, spp not a power of two, T not a multiple of the block, crop window."""
    import mitransient_amd as mitr
    import mitransient_amd.mi as mi
    d = mitr.cornell_box()
    d["integrator"]["amd_mode"] = mode
    d["sensor"]["film"].update(width=40, height=24, temporal_bins=100, start_opl=3.0, bin_width_opl=0.07,
                               crop_width=17, crop_height=9, crop_offset_x=5, crop_offset_y=3)
    scene = mi.load_dict(d)
    s_gpu, t_gpu = gpu_render(scene, 7)
    s_ref, t_ref, *_ = oracle_render(oracle, scene, 7)
    assert t_gpu.shape == (24, 40, 100, 3)
    assert rel_l2(t_gpu, t_ref) <= TOL and rel_l2(s_gpu, s_ref) <= TOL
    # only the crop_size corner of the full-size tensor is written (transient_image_block.py:132,146)
    assert np.all(t_gpu[9:] == 0) and np.all(t_gpu[:, 17:] == 0)


@pytest.mark.parametrize("mode", MODES)
def test_sample_and_pixel_shards_sum_to_whole(oracle, mode):
    """lane identity == RNG identity: sample slices / pixel slices reproduce the full render (multi-GPU basis)."""
    import torch
    scene = make_cornell(width=32, height=32, bins=64, amd_mode=mode)
    s_full, t_full = gpu_render(scene, 12)
    integ = scene.integrator()
    sens = scene.sensors()[0]
    film = sens.film()
    passes = integ.prepare(scene, sens, 0, 12, [])
    for rng in [(0, 5), (5, 6), (6, 12)]:
        integ.accumulate(scene, sens, passes, 12, spp_range=rng)
    s_a, t_a = film.develop()
    assert rel_l2(np.array(t_a), t_full) <= 1e-6 and rel_l2(np.array(s_a), s_full) <= 1e-6
    passes = integ.prepare(scene, sens, 0, 12, [])
    for rng in [(0, 100), (100, 517), (517, 1024)]:
        integ.accumulate(scene, sens, passes, 12, pixel_range=rng)
    s_b, t_b = film.develop()
    assert rel_l2(np.array(t_b), t_full) <= 1e-6 and rel_l2(np.array(s_b), s_full) <= 1e-6
    s_ref, t_ref, *_ = oracle_render(oracle, scene, 12)
    assert rel_l2(np.array(t_b), t_ref) <= TOL


@pytest.mark.parametrize("mode", MODES)
def test_specular_materials(oracle, mode):
    """conductor + dielectric + twosided boxes (the BSDF subset of the north star); in wavefront mode the
    hit queues are sorted by material type (diffuse / conductor / dielectric)."""
    import mitransient_amd as mitr
    import mitransient_amd.mi as mi
    d = mitr.cornell_box()
    d["integrator"]["amd_mode"] = mode
    d["sensor"]["film"].update(width=32, height=32, temporal_bins=128, start_opl=3.0, bin_width_opl=8.0 / 128)
    d["mirror"] = {"type": "conductor", "eta": [0.2, 0.92, 1.1], "k": [3.9, 2.45, 2.14]}
    d["glass"] = {"type": "dielectric", "int_ior": 1.5, "ext_ior": 1.0}
    d["two"] = {"type": "twosided", "bsdf": {"type": "diffuse", "reflectance": {"type": "rgb", "value": [0.2, 0.5, 0.7]}}}
    d["small-box"]["bsdf"] = {"type": "ref", "id": "glass"}
    d["large-box"]["bsdf"] = {"type": "ref", "id": "mirror"}
    d["back"]["bsdf"] = {"type": "ref", "id": "two"}
    d["integrator"]["max_depth"] = 12
    scene = mi.load_dict(d)
    s_gpu, t_gpu = gpu_render(scene, 16)
    s_ref, t_ref, *_ = oracle_render(oracle, scene, 16)
    assert rel_l2(t_gpu, t_ref) <= TOL and rel_l2(s_gpu, s_ref) <= TOL


def test_energy_identity_gpu():
    """transient.sum(axis=2) == steady when the window covers every path
    (examples/transient-nlos/1-simple-nlos-scenes.ipynb, md cell 8)."""
    scene = make_cornell(width=32, height=32, bins=256, start=0.0, window=64.0)
    s_gpu, t_gpu = gpu_render(scene, 32)
    assert rel_l2(t_gpu.sum(axis=2), s_gpu) <= 1e-5


def test_large_film_properties():
    """BASELINE-sized film row (T=1024, 1024 spp) on a strip of pixels: energy identity + determinism
    of the sample set (two renders differ only by f32 summation order)."""
    import torch
    scene = make_cornell(width=512, height=512, bins=1024, start=0.0, window=64.0)
    integ = scene.integrator()
    sens = scene.sensors()[0]
    film = sens.film()
    outs = []
    for _ in range(2):
        passes = integ.prepare(scene, sens, 0, 1024, [])
        integ.accumulate(scene, sens, passes, 1024, pixel_range=(512 * 256, 512 * 256 + 64))
        s, t = film.develop()
        outs.append((np.array(s[256, :64]), np.array(t[256, :64])))
    (s0, t0), (s1, t1) = outs
    assert rel_l2(t0.sum(axis=1), s0) <= 1e-5
    assert rel_l2(t1, t0) <= 1e-6
    assert np.array_equal(t0 != 0, t1 != 0)


# ---------------------------------------------------------------- stand-alone scatter-add
def _splats(n, npix, T, seed=1234, sorted_by_pixel=False):
    rng = np.random.default_rng(seed)
    pixel = rng.integers(0, npix, n).astype(np.uint32)
    if sorted_by_pixel:
        pixel.sort()
    opl = (3.5 + 6.0 * np.clip(rng.normal(400, 120, n), -20, T + 20) / T).astype(np.float32)
    r, g, b = (rng.random(n, dtype=np.float32) for _ in range(3))
    return pixel, opl, r, g, b


@pytest.mark.parametrize("variant,sorted_by_pixel", [(0, False), (0, True), (1, True), (1, False)])      # (1, False): unsorted input -> partitioned by pixel on the device
def test_splat_add_matches_oracle(oracle, variant, sorted_by_pixel):
    import torch
    scene = make_cornell(width=32, height=16, bins=256)
    film = scene.sensors()[0].film()
    film.prepare()
    pixel, opl, r, g, b = _splats(200000, 32 * 16 + 3, 256, sorted_by_pixel=sorted_by_pixel)   # a few ids out of range
    if sorted_by_pixel:
        order = np.argsort(pixel, kind="stable")
        pixel, opl, r, g, b = (x[order] for x in (pixel, opl, r, g, b))
    tt = lambda x: torch.from_numpy(x.view(np.int32) if x.dtype == np.uint32 else x).cuda()
    film.transient_storage.put_opl(tt(pixel), tt(opl), tt(r), tt(g), tt(b), film.desc(), variant)
    torch.cuda.synchronize()
    got = np.array(film.develop(raw=True)[1])
    ref = np.zeros_like(got)
    oracle.splat_add(film.desc(), pixel, opl, r, g, b, ref)
    assert np.array_equal(got != 0, ref != 0)
    assert rel_l2(got, ref) <= TOL


@pytest.mark.parametrize("film_zero", [False, True])
def test_splat_add_sorted_input_with_gaps(oracle, film_zero):
    """mtr_splat_add variant 1 on pixel-sorted input that leaves most pixels EMPTY (the run table is sparse: the pixel behind a
    run holds that run's end and nothing else; the rest of a gap holds nothing) — a lone pixel, gaps of 1, 2 and many pixels,
    an empty first pixel, an empty last pixel."""
    import torch
    from mitransient_amd import _cabi
    W, H, T = 32, 16, 256
    scene = make_cornell(width=W, height=H, bins=T)
    film = scene.sensors()[0].film()
    film.prepare()
    rng = np.random.default_rng(7)
    used = np.array([3, 4, 6, 9, 10, 11, 200, 201, 460, W * H - 2], np.uint32)          # gaps of 0, 1, 2, many; pixels 0 and W*H-1 empty
    pixel = np.sort(used[rng.integers(0, len(used), 20000)]).astype(np.uint32)
    opl = (3.5 + 6.0 * np.clip(rng.normal(400, 120, len(pixel)), -20, T + 20) / T).astype(np.float32)
    r, g, b = (rng.random(len(pixel), dtype=np.float32) for _ in range(3))
    tt = lambda x: torch.from_numpy(x.view(np.int32) if x.dtype == np.uint32 else x).cuda()
    variant = 1 | (_cabi.MTR_SPLAT_FILM_ZERO if film_zero else 0)
    film.transient_storage.put_opl(tt(pixel), tt(opl), tt(r), tt(g), tt(b), film.desc(), variant)
    film.transient_storage.put_opl(tt(pixel[:1]), tt(opl[:1]), tt(r[:1]), tt(g[:1]), tt(b[:1]), film.desc(), 1)      # one record, one pixel
    torch.cuda.synchronize()
    got = np.array(film.develop(raw=True)[1])
    ref = np.zeros_like(got)
    oracle.splat_add(film.desc(), pixel, opl, r, g, b, ref)
    oracle.splat_add(film.desc(), pixel[:1], opl[:1], r[:1], g[:1], b[:1], ref)
    assert np.array_equal(got != 0, ref != 0)
    assert rel_l2(got, ref) <= TOL


@pytest.mark.parametrize("case", ["plain", "film_zero", "odd_sizes", "accumulate", "one_pixel"])
def test_splat_add_partitions_unsorted_input(oracle, case):
    """mtr_splat_add variant 1 on input in ARBITRARY order: the device-side partition by pixel (mtr_splat.hip: two scatter
    passes over 16-byte records) in front of the LDS rows — same film as the oracle's scatter-add, ids out of range and path
    lengths outside the time window dropped, MTR_SPLAT_FILM_ZERO (store-only flush), sizes that are no power of two, a second
    call accumulating onto the first, every contribution in one pixel."""
    import torch
    from mitransient_amd import _cabi
    W, H, T = (32, 16, 256) if case != "odd_sizes" else (37, 11, 301)
    scene = make_cornell(width=W, height=H, bins=T)
    film = scene.sensors()[0].film()
    film.prepare()
    n = 300000 if case != "odd_sizes" else 123457
    pixel, opl, r, g, b = _splats(n, W * H + 3, T, sorted_by_pixel=False)
    if case == "one_pixel":
        pixel[:] = 77
        pixel[::1000] = 5             # (not sorted: 5 after 77)
    tt = lambda x: torch.from_numpy(x.view(np.int32) if x.dtype == np.uint32 else x).cuda()
    variant = 1 | (_cabi.MTR_SPLAT_FILM_ZERO if case == "film_zero" else 0)
    film.transient_storage.put_opl(tt(pixel), tt(opl), tt(r), tt(g), tt(b), film.desc(), variant)
    ref = np.zeros(film.raw_shape(), np.float32)
    oracle.splat_add(film.desc(), pixel, opl, r, g, b, ref)
    if case == "accumulate":          # a second batch lands on the first (read-modify-write flush)
        p2, o2, r2, g2, b2 = _splats(50000, W * H, T, sorted_by_pixel=False)
        p2, o2 = p2[::-1].copy(), o2[::-1].copy()
        film.transient_storage.put_opl(tt(p2), tt(o2), tt(r2), tt(g2), tt(b2), film.desc(), 1)
        oracle.splat_add(film.desc(), p2, o2, r2, g2, b2, ref)
    torch.cuda.synchronize()
    got = np.array(film.develop(raw=True)[1])
    assert np.count_nonzero(ref) > (1000 if case != "one_pixel" else 500)
    assert np.array_equal(got != 0, ref != 0)
    assert rel_l2(got, ref) <= TOL
    assert np.all(got[..., 3] == 0)
    # the workspace the context kept for the partition can be handed back
    from mitransient_amd.runtime import get_context
    ctx = get_context()
    ctx.check(ctx.lib.mtr_ctx_trim(ctx.handle), "mtr_ctx_trim")


@pytest.mark.parametrize("case", ["exhaustive_lasers", "two_pixels", "long_rows", "tile_plus_one", "all_dropped_but_one"])
def test_splat_add_partition_edge_cases(oracle, case):
    """the partition path of mtr_splat_add at its edges: an exhaustive_scan film (row = [laser][t], laser ids out of range
    dropped), a film of two pixels (one bucket, one digit bit per half), rows too long for the fixed-point LDS form (T = 4096:
    the f32 rows), one record more than a staging tile, and input of which a single record survives the time window."""
    import torch
    import mitransient_amd as mitr
    import mitransient_amd.mi as mi
    from mitransient_amd.scene import Properties
    mi.set_variant("llvm_ad_rgb")
    W, H, T, n = {"exhaustive_lasers": (6, 5, 64, 60000), "two_pixels": (2, 1, 8, 5000), "long_rows": (8, 4, 4096, 100000),
                  "tile_plus_one": (32, 16, 256, 4097), "all_dropped_but_one": (16, 16, 64, 20000)}[case]
    props = {"width": W, "height": H, "temporal_bins": T, "start_opl": 3.5, "bin_width_opl": 6.0 / T, "rfilter": {"type": "box"}}
    if case == "exhaustive_lasers":
        props.update(exhaustive_scan=True, laser_scan_width=3, laser_scan_height=2)
    film = mitr.TransientHDRFilm(Properties("transient_hdr_film", props))
    film.prepare([])
    rng = np.random.default_rng(11)
    pixel = rng.integers(0, W * H + (2 if case != "two_pixels" else 0), n).astype(np.uint32)
    opl = (3.5 + 6.0 * np.clip(rng.normal(0.4 * T, 0.12 * T, n), -0.05 * T, 1.05 * T) / T).astype(np.float32)
    if case == "all_dropped_but_one":
        opl[:] = 1.0                      # before the window
        opl[n // 2] = 4.0
    r, g, b = (rng.random(n, dtype=np.float32) for _ in range(3))
    tt = lambda x: torch.from_numpy(x.view(np.int32) if x.dtype == np.uint32 else x).cuda()
    lx = ly = laser = None
    if case == "exhaustive_lasers":
        lx = rng.integers(0, 4, n).astype(np.uint32)          # 3 = out of range
        ly = rng.integers(0, 2, n).astype(np.uint32)
        laser = np.where(lx < 3, lx * 2 + ly, 6).astype(np.uint32)
    assert np.any(np.diff(pixel.astype(np.int64)) < 0)        # unsorted: the partition runs
    film.transient_storage.put_opl(tt(pixel), tt(opl), tt(r), tt(g), tt(b), film.desc(), 1, laser=tt(laser) if laser is not None else None)
    torch.cuda.synchronize()
    got = film.transient_storage.torch_tensor().cpu().numpy()
    ref = np.zeros(film.raw_shape(), np.float32)
    oracle.splat_add(film.desc(), pixel, opl, r, g, b, ref, lx, ly)
    assert np.count_nonzero(ref) > 0
    assert np.array_equal(got != 0, ref != 0)
    assert rel_l2(got, ref) <= TOL
    if case == "all_dropped_but_one":
        assert np.count_nonzero(ref[..., 0]) == 1


def test_splat_add_soak_against_the_contract_form():
    """tools/soak_splat.py, shortened: random film shapes, sizes across tile multiples, orders, zero / non-zero films, the
    workspace reused and trimmed in between — variant 1 (LDS rows, partition) against variant 0 (f32 atomics)"""
    import subprocess, sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak_splat.py"), "3", "12"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "12 calls OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_film_add_transient_data_api(oracle):
    """TransientHDRFilm.add_transient_data with (pos, distance, spec) arrays, incl. the f32 bin-edge KATs."""
    import torch
    scene = make_cornell(width=8, height=8, bins=300, start=3.5, window=6.0)
    film = scene.sensors()[0].film()
    film.bin_width_opl = 0.02
    film.prepare()
    dist = np.array([3.5, 3.5199, 3.52, 3.54, 9.4999, 9.5, 3.4999, np.inf, np.nan], np.float32)
    pos = np.tile(np.array([[2.0, 3.0]], np.float32), (len(dist), 1))
    spec = np.ones((len(dist), 3), np.float32)
    film.add_transient_data(torch.from_numpy(pos), torch.from_numpy(dist), None, torch.from_numpy(spec))
    torch.cuda.synchronize()
    raw = np.array(film.develop(raw=True)[1])
    row = raw[3, 2, :, 0]
    assert row[0] == 3.0 and row[1] == 1.0 and row[299] == 1.0 and row.sum() == 5.0
    assert raw.sum() == 15.0


def test_develop_matches_oracle(oracle):
    import torch
    scene = make_cornell(width=20, height=12, bins=50)
    film = scene.sensors()[0].film()
    film.prepare()
    rng = np.random.default_rng(3)
    t4 = rng.random((12, 20, 50, 4), dtype=np.float32)
    t4[..., 3] = np.where(rng.random((12, 20, 50)) < 0.5, 0.0, t4[..., 3])
    s4 = rng.random((12, 20, 4), dtype=np.float32) * 10
    s4[0, 0, 3] = 0
    film.transient_storage.torch_tensor().copy_(torch.from_numpy(t4))
    film.steady_accum().copy_(torch.from_numpy(s4))
    s, t = film.develop()
    t3, s3 = oracle.develop(film.desc(), t4, s4)
    assert np.array_equal(np.array(t), t3) and np.array_equal(np.array(s), s3)


def test_errors_fail_loudly():
    import mitransient_amd as mitr
    import mitransient_amd.mi as mi
    from mitransient_amd._cabi import MitransientAMDError
    d = mitr.cornell_box()
    d["sensor"]["film"].update(width=16384, height=16384, temporal_bins=4)
    scene = mi.load_dict(d)
    with pytest.raises(Exception, match="film is too big"):
        mi.render(scene, spp=64)              # 2^34 lanes > 2^32 and 2^28 pixels > 2^26 lanes per pass (common.py:62-63)


@pytest.mark.parametrize("mode", MODES)
def test_multi_pass_seeding(oracle, mode):
    """common.py:56-85: a render above the single-pass lane limit is split into passes of floor((2^26-1)/(W*H)) samples, each
    with its own sampler seeded from a seeder sampler (+ a remainder pass); the thresholds are lowered so that a 12 x 10
    film with 11 spp splits into 4 + 4 + 3.  GPU == oracle pass by pass (same seeds, same per-pass lanes, scale 1/11)."""
    import torch
    scene = make_cornell(width=12, height=10, bins=32, amd_mode=mode)
    integ = scene.integrator()
    integ.max_wavefront_size, integ.pass_wavefront_size = 1000, 12 * 10 * 4 + 5
    integ.collect_stats = True
    sens = scene.sensors()[0]
    film = sens.film()
    progress = []
    steady, transient = integ.render(scene, seed=3, spp=11, progress_callback=progress.append)
    torch.cuda.synchronize()
    passes = integ.prepare(scene, sens, 3, 11, [])
    assert [s for _, s in passes] == [4, 4, 3] and len({p.seed_value() for p, _ in passes}) == 3
    assert progress == [pytest.approx(1 / 3), pytest.approx(2 / 3), pytest.approx(1.0)]
    sd = scene.data()
    from oracle import oracle as orc
    bufs = orc.alloc_film(sd.film)
    tot = {k: 0 for k in ("paths", "rays_closest", "rays_shadow", "splats_issued", "bounces")}
    for smp, spp_i in passes:
        _, _, c = orc.render(sd, integ.render_params(film, smp.seed_value(), spp_i, spp_scale=11), use_bvh=True, out=bufs)
        for k in tot:
            tot[k] += c[k]
    t_ref, s_ref = orc.develop(sd.film, bufs[0], bufs[1])
    assert rel_l2(np.array(transient), t_ref) <= TOL and rel_l2(np.array(steady), s_ref) <= TOL
    assert tot["paths"] == 12 * 10 * 11
    # (counters of the GPU render were summed over its three passes)
    steady, transient = integ.render(scene, seed=3, spp=11)
    for k in tot:
        assert integ.total_counters[k] == tot[k], k


def test_wavefront_large_tile_and_overflow(oracle):
    """wavefront mode across several tiles (> 2^20 slots) and with time rows that do not fit LDS
    (T = 16384: every contribution takes the global-atomic path)."""
    scene = make_cornell(width=48, height=48, bins=64, amd_mode="wavefront")
    s_gpu, t_gpu = gpu_render(scene, 600)                       # 1.38 M slots -> 2 tiles, ragged second tile
    s_ref, t_ref, *_ = oracle_render(oracle, scene, 600)
    assert rel_l2(t_gpu, t_ref) <= TOL and rel_l2(s_gpu, s_ref) <= TOL
    scene = make_cornell(width=8, height=8, bins=16384, amd_mode="wavefront")
    s_gpu, t_gpu = gpu_render(scene, 32)
    s_ref, t_ref, *_ = oracle_render(oracle, scene, 32)
    assert rel_l2(t_gpu, t_ref) <= TOL and rel_l2(s_gpu, s_ref) <= TOL
    assert scene.integrator().last_counters["splats_overflow"] > 0
    scene = make_cornell(width=8, height=8, bins=16384, amd_mode="fused")      # fused: row > LDS budget -> HBM atomics
    s_gpu, t_gpu = gpu_render(scene, 32)
    assert rel_l2(t_gpu, t_ref) <= TOL


@pytest.mark.parametrize("mode", MODES)
def test_staircase_like_scene_in_hbm(oracle, mode):
    """BASELINE config-5 stand-in: 852 triangles do not fit the LDS scene budget -> BVH2 node packets and
    triangles are read from HBM/L2, traversal stack depth 16, 7 materials of 3 BSDF types (material-sorted
    queues in wavefront mode), max_depth 65, camera_unwarp."""
    import mitransient_amd.mi as mi
    from mitransient_amd.scenes import staircase_like
    d = staircase_like(n_steps=12, balusters=2, tiles=6, width=40, height=40, temporal_bins=64, spp=8)
    d["integrator"]["amd_mode"] = mode
    scene = mi.load_dict(d)
    s_gpu, t_gpu = gpu_render(scene, 8)
    s_ref, t_ref, s4, t4, cnt = oracle_render(oracle, scene, 8)
    assert rel_l2(t_gpu, t_ref) <= TOL and rel_l2(s_gpu, s_ref) <= TOL
    got = scene.integrator().last_counters
    for k in ("paths", "rays_closest", "rays_shadow", "splats_issued", "bounces"):
        assert got[k] == cnt[k], k


@pytest.mark.parametrize("which", ["lds", "hbm"])
def test_one_sample_per_pixel_in_the_wavefront_organisation(oracle, which):
    """1 spp over a film of more pixels than a segment holds: the segment is cut by its PIXEL count then (k_wf_shade keeps 20 B of
    LDS per pixel of its segment; uncapped, 8192 one-sample pixels asked for 164 KB of LDS and the launch was refused)"""
    import mitransient_amd.mi as mi
    if which == "lds":
        scene = make_cornell(width=128, height=96, bins=32, amd_mode="wavefront")
    else:
        from mitransient_amd.scenes import staircase_like
        d = staircase_like(n_steps=12, balusters=2, tiles=6, width=128, height=96, temporal_bins=32, spp=1)
        d["integrator"]["amd_mode"] = "wavefront"
        scene = mi.load_dict(d)
    s_gpu, t_gpu = gpu_render(scene, 1)
    s_ref, t_ref, s4, t4, cnt = oracle_render(oracle, scene, 1)
    assert rel_l2(t_gpu, t_ref) <= TOL and rel_l2(s_gpu, s_ref) <= TOL
    got = scene.integrator().last_counters
    for k in ("paths", "rays_closest", "rays_shadow", "splats_issued", "bounces"):
        assert got[k] == cnt[k], k


def test_large_procedural_scene_modes_agree():
    """~29k triangles (tiles=120): too slow for the oracle at useful sample counts, so the two independent
    kernel organisations are compared with each other (same lanes, same RNG) + the energy identity."""
    import mitransient_amd.mi as mi
    from mitransient_amd.scenes import staircase_like
    outs = {}
    for mode in MODES:
        d = staircase_like(n_steps=16, balusters=3, tiles=120, width=64, height=64, temporal_bins=128, spp=16)
        d["integrator"]["amd_mode"] = mode
        scene = mi.load_dict(d)
        assert scene.data().tri_verts.shape[0] > 29000
        outs[mode] = gpu_render(scene, 16)
        c = scene.integrator().last_counters
        assert c["paths"] == 64 * 64 * 16
    (s_a, t_a), (s_b, t_b) = outs["fused"], outs["wavefront"]
    assert rel_l2(t_a, t_b) <= TOL and rel_l2(s_a, s_b) <= TOL
    assert rel_l2(t_a.sum(axis=2), s_a) <= 5e-3          # window 0..40 covers nearly every path (a few specular chains run longer)


# ---------------------------------------------------------------- edge cases
def _custom_scene(shapes, film, integrator=None, mode="fused"):
    import mitransient_amd.mi as mi
    from mitransient_amd.transform import ScalarTransform4f as T
    mi.set_variant("llvm_ad_rgb")
    d = {"type": "scene",
         "integrator": dict({"type": "transient_path", "max_depth": 4, "amd_mode": mode}, **(integrator or {})),
         "sensor": {"type": "perspective", "fov": 40.0, "near_clip": 0.01, "far_clip": 50.0,
                    "to_world": T().look_at(origin=[0, 0, 4], target=[0, 0, 0], up=[0, 1, 0]),
                    "sampler": {"type": "independent", "sample_count": 4},
                    "film": dict({"type": "transient_hdr_film", "rfilter": {"type": "box"}}, **film)}}
    d.update(shapes)
    return mi.load_dict(d)


@pytest.mark.parametrize("mode", MODES)
def test_empty_scene_and_unlit_scene(oracle, mode):
    """no geometry at all (every ray escapes) and geometry without any emitter: all-zero outputs, correct counters."""
    from mitransient_amd.transform import ScalarTransform4f as T
    film = {"width": 9, "height": 5, "temporal_bins": 16, "bin_width_opl": 1.0}
    scene = _custom_scene({}, film, mode=mode)
    s, t = gpu_render(scene, 3)
    assert t.shape == (5, 9, 16, 3) and not t.any() and not s.any()
    c = scene.integrator().last_counters
    assert (c["paths"], c["rays_closest"], c["rays_shadow"], c["splats_issued"]) == (135, 135, 0, 0)
    quad = {"wall": {"type": "rectangle", "to_world": T().scale(2.0), "bsdf": {"type": "diffuse", "reflectance": 0.5}}}
    scene = _custom_scene(quad, film, mode=mode)
    s, t = gpu_render(scene, 3)
    s_ref, t_ref, s4, t4, cnt = oracle_render(oracle, scene, 3)
    assert not t.any() and not t_ref.any()
    assert scene.integrator().last_counters["bounces"] == cnt["bounces"]


@pytest.mark.parametrize("mode", MODES)
def test_single_pixel_single_sample_and_emitter_only(oracle, mode):
    """1x1 film, 1 spp, the camera looks straight at a light: the emission lands in exactly one bin."""
    from mitransient_amd.transform import ScalarTransform4f as T
    light = {"light": {"type": "rectangle", "to_world": T().scale(1.0), "bsdf": {"type": "diffuse", "reflectance": 0.0},
                       "emitter": {"type": "area", "radiance": {"type": "rgb", "value": [2.0, 3.0, 4.0]}}}}
    film = {"width": 1, "height": 1, "temporal_bins": 100, "bin_width_opl": 0.1}
    scene = _custom_scene(light, film, mode=mode)
    s, t = gpu_render(scene, 1)
    s_ref, t_ref, *_ = oracle_render(oracle, scene, 1)
    assert np.array_equal(t, t_ref) and np.array_equal(s, s_ref)
    assert np.count_nonzero(t) == 3 and np.allclose(t.sum(axis=(0, 1, 2)), [2.0, 3.0, 4.0])
    assert 39 <= int(np.argmax(t[0, 0, :, 0])) <= 43         # OPL 3.99 on the axis (near clip 0.01) ... 3.99/cos(28 deg) in the corner


def test_many_samples_few_pixels_and_few_samples_many_pixels(oracle):
    """ragged segment shapes of the fused kernel: spp >> lanes per segment, and spp = 1 with hundreds of pixels per segment"""
    for (w, h, spp) in [(3, 2, 3000), (96, 64, 1)]:
        scene = make_cornell(width=w, height=h, bins=32)
        s_gpu, t_gpu = gpu_render(scene, spp)
        s_ref, t_ref, *_ = oracle_render(oracle, scene, spp)
        assert rel_l2(t_gpu, t_ref) <= TOL and rel_l2(s_gpu, s_ref) <= TOL


def test_film_parameters_can_change_between_renders(oracle):
    """mi.traverse(scene)['sensor.film.*'] edits (transient_hdr_film.py:295-308) take effect on the next render."""
    import mitransient_amd.mi as mi
    scene = make_cornell(width=16, height=16, bins=32)
    gpu_render(scene, 4)
    params = mi.traverse(scene)
    params["sensor.film.temporal_bins"] = 80
    params["sensor.film.start_opl"] = 2.0
    params["sensor.film.bin_width_opl"] = 0.125
    params.update()
    s_gpu, t_gpu = gpu_render(scene, 4)
    s_ref, t_ref, *_ = oracle_render(oracle, scene, 4)
    assert t_gpu.shape == (16, 16, 80, 3)
    assert rel_l2(t_gpu, t_ref) <= TOL


def test_two_emitters(oracle):
    """uniform emitter selection with sample reuse [Scene::sample_emitter_direction]"""
    import mitransient_amd as mitr
    import mitransient_amd.mi as mi
    from mitransient_amd.transform import ScalarTransform4f as T
    for mode in MODES:
        d = mitr.cornell_box()
        d["sensor"]["film"].update(width=24, height=24, temporal_bins=64, bin_width_opl=6.0 / 64)
        d["integrator"]["amd_mode"] = mode
        d["light2"] = {"type": "rectangle", "to_world": T().translate([-0.98, 0.0, 0.3]).rotate([0, 1, 0], 90).scale(0.15),
                       "bsdf": {"type": "ref", "id": "white"},
                       "emitter": {"type": "area", "radiance": {"type": "rgb", "value": [4.0, 9.0, 2.0]}}}
        scene = mi.load_dict(d)
        assert scene.data().n_emitters == 2
        s_gpu, t_gpu = gpu_render(scene, 16)
        s_ref, t_ref, *_ = oracle_render(oracle, scene, 16)
        assert rel_l2(t_gpu, t_ref) <= TOL and rel_l2(s_gpu, s_ref) <= TOL


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("with_rect", [False, True])
def test_mesh_area_emitter(oracle, mode, with_rect):
    """area emitter on a triangle mesh (the reference's cbox_diffuse.xml light is an .obj with an `area` emitter):
    [Mesh::sample_position] face pmf with sample reuse + uniform triangle warp, tables in HBM"""
    from test_oracle import mesh_light_cornell
    scene = mesh_light_cornell(width=32, height=32, bins=128, with_rect=with_rect)
    scene.integrator().amd_mode = mode
    s_gpu, t_gpu = gpu_render(scene, 16, seed=5)
    s_ref, t_ref, s4, t4, cnt = oracle_render(oracle, scene, 16, seed=5)
    assert rel_l2(t_gpu, t_ref) <= TOL and rel_l2(s_gpu, s_ref) <= TOL
    got = scene.integrator().last_counters
    for k in ("paths", "rays_closest", "rays_shadow", "splats_issued", "bounces"):
        assert got[k] == cnt[k], k


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("name", ["cbox_diffuse", "cbox_mirror"])
def test_reference_example_scenes(oracle, name, mode):
    """the reference's examples/transient/cornell-box/{cbox_diffuse,cbox_mirror}.xml (flattened fixtures; the
    reference's units: box 550 wide, near clip 10, 400 bins of 6.5 from OPL 1000; mesh light, conductor + glass)"""
    import os
    from mitransient_amd.scenes import from_fixture
    GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    scene = from_fixture(os.path.join(GOLDEN_DIR, f"{name}_scene.npz"), film={"width": 48, "height": 48},
                         integrator={"amd_mode": mode}, spp=32)
    s_gpu, t_gpu = gpu_render(scene, 32)
    s_ref, t_ref, s4, t4, cnt = oracle_render(oracle, scene, 32)
    assert t_gpu.shape == (48, 48, 400, 3)
    assert rel_l2(t_gpu, t_ref) <= TOL and rel_l2(s_gpu, s_ref) <= TOL
    got = scene.integrator().last_counters
    for k in ("paths", "rays_closest", "rays_shadow", "splats_issued", "bounces"):
        assert got[k] == cnt[k], k


@pytest.mark.parametrize("mode", MODES)
def test_staircase_config5_geometry(oracle, mode):
    """BASELINE config 5 geometry: examples/diff-transient/staircase/scene.xml, 262,663 triangles in HBM,
    max_depth 65, camera_unwarp (approximate materials), reduced film"""
    from mitransient_amd.scenes import staircase
    scene = staircase(width=45, height=80, spp=4, amd_mode=mode)
    s_gpu, t_gpu = gpu_render(scene, 4)
    s_ref, t_ref, s4, t4, cnt = oracle_render(oracle, scene, 4)
    assert rel_l2(t_gpu, t_ref) <= TOL and rel_l2(s_gpu, s_ref) <= TOL
    got = scene.integrator().last_counters
    for k in ("paths", "rays_closest", "rays_shadow", "splats_issued", "bounces"):
        assert got[k] == cnt[k], k


@pytest.mark.parametrize("max_depth,rr_depth,unwarp", [(1, 5, False), (2, 5, True), (3, 1, False), (-1, 2, True)])
def test_hbm_scene_wavefront_depths(oracle, max_depth, rr_depth, unwarp):
    """the wavefront pipeline of scenes in HBM/L2 (k_wf_trace closest -> k_wf_shadow_gen -> k_wf_trace any-hit ->
    k_wf_shade) at the depth edge cases: no emitter sampling at all (max_depth 1), one bounce, early Russian
    roulette, unbounded depth (host polls the live count); ragged film so that the last segment is partial"""
    import mitransient_amd.mi as mi
    from mitransient_amd.scenes import staircase_like
    d = staircase_like(n_steps=12, balusters=2, tiles=6, width=37, height=23, temporal_bins=48, spp=6, max_depth=max_depth)
    d["integrator"].update(amd_mode="wavefront", rr_depth=rr_depth, camera_unwarp=unwarp)
    scene = mi.load_dict(d)
    s_gpu, t_gpu = gpu_render(scene, 6, seed=3)
    s_ref, t_ref, s4, t4, cnt = oracle_render(oracle, scene, 6, seed=3)
    assert rel_l2(t_gpu, t_ref) <= TOL and rel_l2(s_gpu, s_ref) <= TOL
    got = scene.integrator().last_counters
    for k in ("paths", "rays_closest", "rays_shadow", "splats_issued", "bounces"):
        assert got[k] == cnt[k], k
    if max_depth == 1:
        assert cnt["rays_shadow"] == 0


@pytest.mark.parametrize("bins,spp", [(1024, 40), (2048, 300), (4096, 520), (4100, 64), (12000, 33)])
def test_fused_row_ring_sizes(oracle, bins, spp):
    """k_fused's ring of row histograms at every depth the LDS budget produces: 4 / 2 / 1 slots (T = 1024 / 2048 / 4096:
    the last one runs the 3-waves-per-SIMD instantiation with the 4-bins-per-lane flush), a row that is not a multiple of
    four bins, and the longest row that still fits a CU on its own (T = 12000); samples per pixel below, around and above
    the 256 lanes of a workgroup; a second pass accumulated onto the first (read-modify-write flush)."""
    import torch
    scene = make_cornell(width=12, height=10, bins=bins, start=3.0, window=9.0, amd_mode="fused")
    s_gpu, t_gpu = gpu_render(scene, spp)
    s_ref, t_ref, s4, t4, cnt = oracle_render(oracle, scene, spp)
    assert rel_l2(t_gpu, t_ref) <= TOL and rel_l2(s_gpu, s_ref) <= TOL
    got = scene.integrator().last_counters
    for k in ("paths", "rays_closest", "rays_shadow", "splats_issued", "bounces"):
        assert got[k] == cnt[k], k
    # two half passes accumulated into one film == one pass over all samples (the second pass finds a non-zero film)
    integ = scene.integrator()
    sens = scene.sensors()[0]
    film = sens.film()
    passes = integ.prepare(scene, sens, 0, spp, integ.aov_names())
    half = spp // 2
    integ.accumulate(scene, sens, passes, spp, spp_range=(0, half))
    integ.accumulate(scene, sens, passes, spp, spp_range=(half, spp))
    s2, t2 = film.develop()
    torch.cuda.synchronize()
    assert rel_l2(np.array(t2), t_ref) <= TOL and rel_l2(np.array(s2), s_ref) <= TOL


def test_warn_negative_and_invalid_sample_values(caplog):
    """TransientImageBlock.put_ (transient_image_block.py:107-125): with warn_negative / warn_invalid a bad value among the
    active samples is logged as 'Invalid sample value: [...]' (and still accumulated, as in the reference)"""
    import logging
    import torch
    from mitransient_amd.render.transient_image_block import TransientImageBlock
    from mitransient_amd import _cabi
    fd = _cabi.mtr_film_desc()
    fd.width = fd.crop_width = 4
    fd.height = fd.crop_height = 2
    fd.temporal_bins, fd.start_opl, fd.bin_width_opl = 8, 0.0, 1.0
    blk = TransientImageBlock(size_xyt=(4, 2, 8), warn_negative=True, warn_invalid=True)
    pix = torch.tensor([0, 1, 2], dtype=torch.int32, device="cuda")
    opl = torch.tensor([0.5, 1.5, 2.5], device="cuda")
    ok = torch.tensor([1.0, 2.0, 3.0], device="cuda")
    with caplog.at_level(logging.WARNING, logger="mitransient_amd"):
        blk.put_opl(pix, opl, ok, ok, ok, fd)
        assert not caplog.records
        blk.put_opl(pix, opl, torch.tensor([1.0, -0.5, 3.0], device="cuda"), ok, ok, fd)
        blk.put_opl(pix, opl, ok, torch.tensor([1.0, 2.0, float("nan")], device="cuda"), ok, fd)
    assert len(caplog.records) == 2 and all("Invalid sample value" in r.getMessage() for r in caplog.records)
    t = blk.torch_tensor().cpu().numpy()
    assert t[0, 0, 0, 0] == 3.0 and t[0, 1, 1, 0] == 3.5


@pytest.mark.parametrize("bins", [256, 1024, 4096])
def test_deterministic_fused_rows(oracle, bins):
    """amd_deterministic (MTR_FLAG_DETERMINISTIC): the fused kernel's LDS rows and steady sums in 64-bit fixed point — sums no
    longer depend on the order in which lanes add, so two renders are bit for bit equal (f32 LDS atomics: equal only up to
    summation order); still the oracle's film within 1e-5, identical counters.  Three row lengths = three ring depths."""
    scene = make_cornell(width=20, height=12, bins=bins, start=3.0, window=9.0, amd_mode="fused", amd_deterministic=True)
    s0, t0 = gpu_render(scene, 200)
    c0 = dict(scene.integrator().last_counters)
    s1, t1 = gpu_render(scene, 200)
    assert np.array_equal(t0, t1) and np.array_equal(s0, s1)
    s_ref, t_ref, s4, t4, cnt = oracle_render(oracle, scene, 200)
    assert rel_l2(t0, t_ref) <= TOL and rel_l2(s0, s_ref) <= TOL
    assert np.array_equal(t0 != 0, t_ref != 0)
    for k in ("paths", "rays_closest", "rays_shadow", "splats_issued", "bounces"):
        assert c0[k] == cnt[k], k
    # two passes accumulated onto each other stay reproducible too (the flush reads, adds, writes in pass order)
    integ, sens = scene.integrator(), scene.sensors()[0]
    outs = []
    for _ in range(2):
        passes = integ.prepare(scene, sens, 0, 200, [])
        integ.accumulate(scene, sens, passes, 200, spp_range=(0, 77))
        integ.accumulate(scene, sens, passes, 200, spp_range=(77, 200))
        s, t = sens.film().develop()
        outs.append(np.array(t))
    assert np.array_equal(outs[0], outs[1]) and rel_l2(outs[0], t_ref) <= TOL


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("which", ["flip+turn", "floor"])
def test_flip_normals_on_rectangles(oracle, mode, which):
    """`flip_normals` on analytic rectangles (an emitter turned away and flipped back; a flipped one-sided floor)"""
    from test_oracle import _flipped_cornell
    scene = _flipped_cornell("flip+turn" if which == "flip+turn" else None, floor_flipped=(which == "floor"))
    scene.integrator().amd_mode = mode
    s_gpu, t_gpu = gpu_render(scene, 16, seed=2)
    s_ref, t_ref, s4, t4, cnt = oracle_render(oracle, scene, 16, seed=2)
    assert rel_l2(t_gpu, t_ref) <= TOL and rel_l2(s_gpu, s_ref) <= TOL
    got = scene.integrator().last_counters
    for k in ("paths", "rays_closest", "rays_shadow", "splats_issued", "bounces"):
        assert got[k] == cnt[k], k


@pytest.mark.parametrize("det", [False, True], ids=["f32-rows", "fixed-point-rows"])
def test_single_pass_film_lifecycle(oracle, det):
    """MTR_FLAG_DEVELOPED_ROWS: one pass of the fused kernel stores the developed (H,W,T,3) rows whole — no cleared block,
    no develop pass — and gives what clear + accumulate + develop gives (bit for bit with order-independent rows)."""
    import torch
    a = make_cornell(width=40, height=24, bins=100, amd_mode="fused", amd_deterministic=det)
    b = make_cornell(width=40, height=24, bins=100, amd_mode="fused", amd_deterministic=det, amd_direct_develop=False)
    fa, fb = a.sensors()[0].film(), b.sensors()[0].film()
    # the output tensor starts as garbage: every row must be written, zeros included
    sa, ta = gpu_render(a, 16, seed=4)
    assert fa.direct_develop and fa.transient_storage is None and fa.developed_storage() is not None
    fa.developed_storage().fill_(float("nan"))
    sa, ta = gpu_render(a, 16, seed=4)
    sb, tb = gpu_render(b, 16, seed=4)
    assert not fb.direct_develop and fb.transient_storage is not None
    assert ta.shape == tb.shape == (24, 40, 100, 3) and np.isfinite(ta).all()
    if det:
        assert np.array_equal(ta, tb)
    assert rel_l2(ta, tb) <= 1e-6 and rel_l2(sa, sb) <= 1e-6
    s_ref, t_ref, s4, t4, cnt = oracle_render(oracle, a, 16, seed=4)
    assert rel_l2(ta, t_ref) <= TOL and a.integrator().last_counters["splats_issued"] == cnt["splats_issued"]
    # the raw block on request: (developed rgb, weight 0)
    s_raw, t_raw = fa.develop(raw=True)
    t_raw = np.array(t_raw)
    assert t_raw.shape == (24, 40, 100, 4) and np.array_equal(t_raw[..., :3], ta) and np.all(t_raw[..., 3] == 0)
    # a render in parts (sample shards) on the same film falls back to the accumulating block
    integ = a.integrator()
    passes = integ.prepare(a, a.sensors()[0], 4, 16, [])
    assert not fa.direct_develop
    integ.accumulate(a, a.sensors()[0], passes, 16, spp_range=(0, 7))
    integ.accumulate(a, a.sensors()[0], passes, 16, spp_range=(7, 16))
    torch.cuda.synchronize()
    assert rel_l2(np.array(fa.develop()[1]), tb) <= 1e-6
    # a crop window, or the wavefront organisation, keep the classic lifecycle
    c = make_cornell(width=40, height=24, bins=100, amd_mode="wavefront")
    gpu_render(c, 4)
    assert not c.sensors()[0].film().direct_develop


@pytest.mark.gpu
def test_results_of_consecutive_renders_stay_valid(oracle):
    """develop() of the single-pass lifecycle returns the film's own tensor without a copy — so it must give that tensor UP:
    a result the caller still holds on the device is not overwritten by the next render (the reference returns a fresh tensor
    from every develop()).  ADVICE round 3: (a + b) / 2 of two renders silently became b."""
    import torch
    import mitransient_amd.mi as mi
    a = make_cornell(width=40, height=24, bins=100, amd_mode="fused")
    film = a.sensors()[0].film()
    s0, t0 = mi.render(a, spp=16, seed=0)
    assert film.direct_develop
    keep0 = t0.torch().clone()
    s1, t1 = mi.render(a, spp=16, seed=1)
    torch.cuda.synchronize()
    assert t0.torch().data_ptr() != t1.torch().data_ptr()
    assert torch.equal(t0.torch(), keep0)                     # the first result is what it was
    assert not torch.equal(t0.torch(), t1.torch())            # ... and the second is another render
    mean = ((t0.torch() + t1.torch()) / 2).cpu().numpy()
    r0 = oracle_render(oracle, a, 16, seed=0)[1]
    r1 = oracle_render(oracle, a, 16, seed=1)[1]
    assert rel_l2(mean, (r0 + r1) / 2) <= TOL
    # a result that was dropped gives its memory back: the third render needs no third tensor
    del t0, keep0
    s2, t2 = mi.render(a, spp=16, seed=2)
    torch.cuda.synchronize()
    assert rel_l2(np.array(t2), oracle_render(oracle, a, 16, seed=2)[1]) <= TOL
    assert torch.equal(t1.torch(), torch.as_tensor(np.array(t1), device=t1.torch().device))


@pytest.mark.gpu
def test_fallbacks_after_a_direct_prepare_start_from_zero(oracle):
    """prepare() of the single-pass lifecycle leaves the developed tensor UNINITIALISED (the render stores every row).  Whatever
    then accumulates in parts instead — sample shards, contributions from Python — must start from a zeroed block, not from a
    copy of that garbage."""
    import torch
    a = make_cornell(width=40, height=24, bins=100, amd_mode="fused")
    sens = a.sensors()[0]
    film, integ = sens.film(), a.integrator()
    t_ref = oracle_render(oracle, a, 16, seed=4)[1]
    for how in ("shards", "python"):
        passes = integ.prepare(a, sens, 4, 16, [], _direct_develop=True)
        assert film.direct_develop and film.transient_storage is None
        film.developed_storage().fill_(float("nan"))           # what torch.empty may hold
        if how == "shards":
            integ.accumulate(a, sens, passes, 16, spp_range=(0, 7))
            integ.accumulate(a, sens, passes, 16, spp_range=(7, 16))
            torch.cuda.synchronize()
            got = np.array(film.develop()[1])
            assert np.isfinite(got).all() and rel_l2(got, t_ref) <= TOL
        else:
            pos = torch.tensor([[3.5, 2.5], [3.5, 2.5], [10.2, 7.9]], device="cuda")
            dist = torch.tensor([4.0, 4.0, 5.0], device="cuda")
            spec = torch.tensor([[1.0, 2.0, 3.0], [0.5, 0.5, 0.5], [4.0, 0.0, 1.0]], device="cuda")
            film.add_transient_data(pos, dist, None, spec)
            torch.cuda.synchronize()
            got = np.array(film.develop()[1])
            want = np.zeros_like(got)
            b4, b5 = int((4.0 - 3.5) / (6.0 / 100)), int((5.0 - 3.5) / (6.0 / 100))
            want[2, 3, b4] = [1.5, 2.5, 3.5]
            want[7, 10, b5] = [4.0, 0.0, 1.0]
            assert np.isfinite(got).all() and np.allclose(got, want, rtol=1e-6, atol=0)
    # develop() before any render of a direct film is an error, not garbage
    integ.prepare(a, sens, 4, 16, [], _direct_develop=True)
    with pytest.raises(RuntimeError):
        film.develop()


@pytest.mark.parametrize("seeding", ["tea", "tea+lane", "tea64"])
def test_sampler_seeding_variants(oracle, seeding):
    """the three readings of mitsuba's PCG32Sampler::seed (MTR_FLAG_PCG_INITSEQ_PLUS_LANE, MTR_FLAG_PCG_TEA64): kernels and oracle
    draw the same streams under each, and the streams differ from one another (tests/test_reference_golden.py decides between
    them the day a real reference render exists)"""
    outs = {}
    for mode in MODES:
        scene = make_cornell(width=32, height=32, bins=64, amd_mode=mode)
        integ = scene.integrator()
        integ.pcg_initseq_plus_lane = seeding == "tea+lane"
        integ.pcg_tea64 = seeding == "tea64"
        s_gpu, t_gpu = gpu_render(scene, 8)
        s_ref, t_ref, s4, t4, cnt = oracle_render(oracle, scene, 8)
        assert rel_l2(t_gpu, t_ref) <= TOL and rel_l2(s_gpu, s_ref) <= TOL
        got = integ.last_counters
        for k in ("paths", "rays_closest", "rays_shadow", "splats_issued"):
            assert got[k] == cnt[k], k
        outs[mode] = t_gpu
    plain = make_cornell(width=32, height=32, bins=64)
    _, t_plain = gpu_render(plain, 8)
    assert (seeding == "tea") == (rel_l2(outs["fused"], t_plain) <= TOL)


def test_auto_mode_follows_the_depth_of_the_tree(oracle, tmp_path, monkeypatch):
    """MTR_MODE_AUTO (mtr_render_plan): the fused kernel for a scene staged in LDS whose 8-wide tree is shallow (the Cornell box: root +
    object nodes), the wavefront organisation as soon as the tree is deeper (the same box with its two boxes as 2 x 2-tessellated
    meshes, 108 triangles: 204 ms fused against 125 ms at config 2's size — profiles/r05_size_sweep.txt); both agree with the oracle"""
    import sys
    import mitransient_amd.mi as mi
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import size_sweep
    monkeypatch.setattr(size_sweep, "TMP", str(tmp_path))
    picked = {}
    for n in (1, 2):
        scene = mi.load_dict(size_sweep.cornell(n, 48, 48, 96))
        s_gpu, t_gpu = gpu_render(scene, 8)
        s_ref, t_ref, *_ = oracle_render(oracle, scene, 8)
        assert rel_l2(t_gpu, t_ref) <= TOL and rel_l2(s_gpu, s_ref) <= TOL
        picked[n] = "wavefront" if scene.integrator().last_times["scatter_launches"] else "fused"
    assert picked == {1: "fused", 2: "wavefront"}, picked


def test_specialised_kernels_return_the_bits_of_the_general_ones(oracle):
    """scene traits (mtr_core.h kTr*): the Cornell box runs the kernels specialised on 'diffuse materials, one rectangle emitter,
    leaves of one pair'; the same box with a SECOND, black rectangle emitter far outside runs the general ones.  A black emitter adds
    nothing to any sum but halves every emitter-sampling pdf, so radiance cannot be compared — the traversal can: the time-bin
    cells touched by direct emission (max_depth 1: no emitter sampling at all) must be identical, bit for bit."""
    import mitransient_amd as mitr
    import mitransient_amd.mi as mi
    from mitransient_amd.transform import ScalarTransform4f as T
    mi.set_variant("llvm_ad_rgb")
    outs = []
    for extra in (False, True):
        d = mitr.cornell_box()
        d["sensor"]["film"].update(width=48, height=48, temporal_bins=96, start_opl=3.5, bin_width_opl=6.0 / 96)
        d["integrator"]["max_depth"] = 1
        d["integrator"]["amd_mode"] = "fused"
        if extra:
            d["far-light"] = {"type": "rectangle", "to_world": T().translate([50.0, 50.0, 50.0]).scale([0.01, 0.01, 0.01]),
                              "bsdf": {"type": "ref", "id": "white"},
                              "emitter": {"type": "area", "radiance": {"type": "rgb", "value": [0.0, 0.0, 0.0]}}}
        scene = mi.load_dict(d)
        outs.append(gpu_render(scene, 16)[1])
    assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32))


def test_flat_top_level_returns_the_bits_of_the_tree_walk(oracle):
    """scene trait kTrFlatTop (mtr_core.h flat_walk_device): the Cornell box — rectangles and box nodes under one root — is not
    walked at all; the same box with one more shape, a small triangle BEHIND the closed back wall that no ray can reach, has a
    triangle leaf at its top level and is walked through its tree.  Neither the primitive tests nor the tie rule differ, only what is
    culled: with order-independent film rows (amd_deterministic) the two films are equal bit for bit, and so are the counters."""
    import mitransient_amd as mitr
    import mitransient_amd.mi as mi
    from mitransient_amd import _cabi
    mi.set_variant("llvm_ad_rgb")
    outs, cnts, flat = [], [], []
    for extra in (False, True):
        d = mitr.cornell_box()
        d["sensor"]["film"].update(width=40, height=40, temporal_bins=128, start_opl=3.5, bin_width_opl=6.0 / 128)
        d["integrator"].update(amd_mode="fused", amd_deterministic=True)
        if extra:
            tri = os.path.join(ROOT, "tests", "_build", "far_triangle.obj")
            os.makedirs(os.path.dirname(tri), exist_ok=True)
            with open(tri, "w") as fh:
                fh.write("v -0.1 -0.1 -30\nv 0.1 -0.1 -30\nv 0 0.1 -30\nf 1 2 3\n")
            d["far-triangle"] = {"type": "obj", "filename": tri, "face_normals": True, "bsdf": {"type": "ref", "id": "white"}}
        scene = mi.load_dict(d)
        s, t = gpu_render(scene, 64, seed=7)
        outs.append((s, t)); cnts.append(dict(scene.integrator().last_counters))
        flat.append(bool(scene.gpu_traits() & _cabi.MTR_TRAIT_FLAT_TOP))
    assert flat == [True, False]
    assert np.array_equal(outs[0][1].view(np.uint32), outs[1][1].view(np.uint32))
    assert np.array_equal(outs[0][0].view(np.uint32), outs[1][0].view(np.uint32))
    for k in ("paths", "rays_closest", "rays_shadow", "splats_issued", "bounces"):
        assert cnts[0][k] == cnts[1][k], k
    s_ref, t_ref, s4, t4, cnt = oracle_render(oracle, scene, 64, seed=7)
    assert rel_l2(outs[0][1], t_ref) <= TOL and cnts[0]["rays_shadow"] == cnt["rays_shadow"]


@pytest.mark.parametrize("case", ["seven-rects-one-cube", "four-rects-four-cubes", "touching-and-nested-cubes", "camera-inside-a-cube", "one-cube-one-rect", "five-cubes",
                                  "rects-cube-and-loose-triangles"])
def test_flat_top_level_variants(oracle, case):
    """flat_walk_device beyond the Cornell box: an odd number of rectangles (7 + 1 cube: the fourth slab pair and its masked second
    half), the most boxes the flat top level takes (4, under 4 rectangles: the root holds 8 children), cubes that touch along a face and one nested in another (two boxes' faces
    at one distance: the tie rule), the camera INSIDE a cube (box_select's 'origin inside' branch for every camera ray), the
    smallest flat scene, one box too many (the tree walk again), and triangle leaves at the top level (a two-triangle mesh: the rectangle
    stage tests rectangles and pair leaves alike, under the general shading code).  Film to 1e-5, counters equal, against the oracle."""
    import mitransient_amd as mitr
    import mitransient_amd.mi as mi
    from mitransient_amd import _cabi
    from mitransient_amd.transform import ScalarTransform4f as T
    mi.set_variant("llvm_ad_rgb")
    d = mitr.cornell_box()
    d["sensor"]["film"].update(width=32, height=32, temporal_bins=96, start_opl=0.0, bin_width_opl=12.0 / 96)
    d["integrator"].update(amd_mode="fused", max_depth=6)
    white = {"type": "ref", "id": "white"}

    def cube(at, deg, scale):
        return {"type": "cube", "to_world": T().translate(at).rotate([0, 1, 0], deg).scale(scale), "bsdf": white}
    flat = True
    if case == "seven-rects-one-cube":
        d.pop("large-box")
        d["shelf"] = {"type": "rectangle", "to_world": T().translate([0.0, 0.1, -0.6]).rotate([1, 0, 0], -70.0).scale([0.5, 0.2, 1.0]), "bsdf": {"type": "ref", "id": "green"}}
    elif case == "four-rects-four-cubes":
        d.pop("green-wall"); d.pop("red-wall")
        d["third-box"] = cube([0.5, 0.3, -0.5], 31.0, [0.15, 0.2, 0.1])
        d["fourth-box"] = cube([-0.6, -0.8, 0.5], -40.0, 0.18)
    elif case == "touching-and-nested-cubes":
        d.pop("green-wall"); d.pop("red-wall")
        d["small-box"] = cube([0.3, -0.7, 0.3], 0.0, 0.3)
        d["large-box"] = cube([-0.3, -0.7, 0.3], 0.0, 0.3)                  # shares the plane x = 0 with the other
        d["inner-box"] = cube([0.3, -0.7, 0.3], 0.0, 0.15)                   # wholly inside the first
        d["flush-box"] = cube([0.3, -0.25, 0.3], 0.0, [0.3, 0.15, 0.3])      # stands on the first: two coincident faces
    elif case == "camera-inside-a-cube":
        d["large-box"] = cube([0.0, 0.0, 3.9], 10.0, [0.5, 0.5, 0.5])        # around the camera; its inside faces face away: rays leave through them
    elif case == "one-cube-one-rect":
        for k in ("floor", "ceiling", "back", "green-wall", "red-wall", "small-box"):
            d.pop(k)
    elif case == "rects-cube-and-loose-triangles":          # triangle leaves at the top level, beside rectangles and a box node
        d.pop("small-box")
        tri = os.path.join(ROOT, "tests", "_build", "kite.obj")
        os.makedirs(os.path.dirname(tri), exist_ok=True)
        with open(tri, "w") as fh:
            fh.write("v 0.2 -0.9 0.6\nv 0.7 -0.9 0.2\nv 0.5 -0.2 0.4\nv 0.1 -0.3 0.1\nf 1 2 3\nf 1 3 4\n")
        d["kite"] = {"type": "obj", "filename": tri, "face_normals": True, "bsdf": {"type": "twosided", "bsdf": {"type": "diffuse", "reflectance": {"type": "rgb", "value": [0.4, 0.5, 0.6]}}}}
    else:
        for i in range(3):
            d[f"extra-box-{i}"] = cube([-0.6 + 0.6 * i, 0.5, -0.6], 15.0 * i, 0.12)
        flat = False
    scene = mi.load_dict(d)
    assert bool(scene.gpu_traits() & _cabi.MTR_TRAIT_FLAT_TOP) == flat
    s_gpu, t_gpu = gpu_render(scene, 24, seed=11)
    s_ref, t_ref, s4, t4, cnt = oracle_render(oracle, scene, 24, seed=11)
    assert rel_l2(t_gpu, t_ref) <= TOL and rel_l2(s_gpu, s_ref) <= TOL
    got = scene.integrator().last_counters
    for k in ("paths", "rays_closest", "rays_shadow", "splats_issued", "bounces"):
        assert got[k] == cnt[k], k
