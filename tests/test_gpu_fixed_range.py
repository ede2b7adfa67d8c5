"""The reference's film is plain f32 (`render/transient_image_block.py:79-81`: `dr.scatter_reduce(Add)` — values of any size,
Inf and NaN propagate).  The kernels sum film rows in signed 2^-42 fixed point (range +-2^21) wherever the row fits LDS:
these tests drive values of 1e7, +-Inf, NaN and bin sums beyond 2^21 through every kernel that does so — `k_wf_scatter`,
`mtr_splat_add` variant 1 (sorted input: `k_splat_rows`; arbitrary order: the partition + `k_splat_rows_rec`) and the
`amd_deterministic` rows of `k_fused` — and compare the film with the oracle's f32 film CELL FOR CELL.
"""
import numpy as np
import pytest

from conftest import make_cornell, rel_l2
from test_gpu_parity import gpu_render, oracle_render

pytestmark = pytest.mark.gpu


def assert_cells_equal(got, ref, rtol=2e-5):
    """cell for cell: the same NaNs, the same infinities, finite cells within rtol of the cell (f32 summation order)"""
    got, ref = np.asarray(got), np.asarray(ref)
    assert got.shape == ref.shape
    assert np.array_equal(np.isnan(got), np.isnan(ref)), "NaN cells differ"
    inf = np.isinf(ref)
    assert np.array_equal(np.isinf(got), inf) and np.array_equal(got[inf], ref[inf]), "infinite cells differ"
    fin = np.isfinite(ref)
    assert np.array_equal(got[fin] != 0, ref[fin] != 0), "touched cells differ"
    assert np.allclose(got[fin], ref[fin], rtol=rtol, atol=0.0), float(np.max(np.abs(got[fin] - ref[fin]) / np.maximum(np.abs(ref[fin]), 1e-30)))


def _wild_splats(n, npix, T, seed, sorted_by_pixel):
    """mostly U(0,1); per pixel class: 1e7-sized values, a bin whose SUM leaves 2^21 although no addend does, +Inf, -Inf, NaN.
    Large addends of one cell share a sign (exact cancellations of 1e7 would make the cell depend on the summation order)."""
    rng = np.random.default_rng(seed)
    pixel = rng.integers(0, npix, n).astype(np.uint32)
    opl = (3.5 + 6.0 * np.clip(rng.normal(400, 120, n), -20, T + 20) / T).astype(np.float32)
    r, g, b = (rng.random(n, dtype=np.float32) for _ in range(3))
    cls = pixel % 16
    pick = rng.random(n) < 0.02
    big = pick & (cls == 1); r[big] = 1.0e7; g[big] = 3.3e6
    neg = pick & (cls == 2); b[neg] = -2.5e7
    pinf = pick & (cls == 3); r[pinf] = np.inf
    ninf = pick & (cls == 4); g[ninf] = -np.inf
    nan = pick & (cls == 5); b[nan] = np.nan
    both = pick & (cls == 6); r[both] = np.where(rng.random(int(both.sum())) < 0.5, np.inf, -np.inf)      # +Inf and -Inf may meet: NaN
    # a bin sum beyond 2^21 from addends far below 2^20: every contribution of these pixels is 3000 in one bin
    heap = cls == 7
    r[heap] = 3000.0; g[heap] = 2999.0; b[heap] = 0.25; opl[heap] = np.float32(3.5 + 6.0 * 100.5 / T)
    if sorted_by_pixel:
        order = np.argsort(pixel, kind="stable")
        pixel, opl, r, g, b = (x[order] for x in (pixel, opl, r, g, b))
    return pixel, opl, r, g, b


@pytest.mark.parametrize("case", ["sorted", "sorted_film_zero", "arbitrary_order", "arbitrary_order_film_zero", "contract_form"])
def test_splat_add_values_beyond_the_fixed_point_range(oracle, case):
    import torch
    from mitransient_amd import _cabi
    W, H, T = 32, 16, 256
    scene = make_cornell(width=W, height=H, bins=T)
    film = scene.sensors()[0].film()
    film.prepare()
    srt = case.startswith("sorted")
    pixel, opl, r, g, b = _wild_splats(600000, W * H, T, 99, srt)
    assert int((pixel % 16 == 7).sum()) * 3000.0 / (W * H / 16) > 2 ** 21          # the heaps do leave the fixed-point range
    tt = lambda x: torch.from_numpy(x.view(np.int32) if x.dtype == np.uint32 else x).cuda()
    variant = 0 if case == "contract_form" else 1
    if case.endswith("film_zero"):
        variant |= _cabi.MTR_SPLAT_FILM_ZERO
    film.transient_storage.put_opl(tt(pixel), tt(opl), tt(r), tt(g), tt(b), film.desc(), variant)
    torch.cuda.synchronize()
    got = np.array(film.develop(raw=True)[1])
    ref = np.zeros_like(got)
    oracle.splat_add(film.desc(), pixel, opl, r, g, b, ref)
    assert np.isnan(ref).any() and np.isinf(ref).any() and np.nanmax(np.where(np.isfinite(ref), ref, 0)) > 2 ** 21
    assert_cells_equal(got, ref)
    # and once more onto the same film (the flush reads, adds, writes): finite cells double, the others stay what they are
    if not case.endswith("film_zero"):
        film.transient_storage.put_opl(tt(pixel), tt(opl), tt(r), tt(g), tt(b), film.desc(), variant)
        torch.cuda.synchronize()
        oracle.splat_add(film.desc(), pixel, opl, r, g, b, ref)
        assert_cells_equal(np.array(film.develop(raw=True)[1]), ref)


def _bright_cornell(radiance, **kw):
    import mitransient_amd as mitr
    import mitransient_amd.mi as mi
    mi.set_variant("llvm_ad_rgb")
    d = mitr.cornell_box()
    d["sensor"]["film"].update(width=24, height=16, temporal_bins=128, start_opl=3.0, bin_width_opl=9.0 / 128)
    d["light"]["emitter"]["radiance"] = dict(type="rgb", value=list(radiance))
    d["integrator"].update(kw)
    return mi.load_dict(d)


@pytest.mark.parametrize("org", ["wavefront", "fused-deterministic", "fused"])
@pytest.mark.parametrize("radiance", [(1.8e10, 1.4e10, 6.7e9), (float("inf"), 13.9, 6.7), (18.3, float("nan"), 6.7)],
                         ids=["1e10", "inf", "nan"])
def test_render_radiance_beyond_the_fixed_point_range(oracle, org, radiance):
    """an emitter of radiance 1e10 puts contributions of 1e7 and bin sums of 1e8 on the film; an infinite or NaN channel puts
    Inf / NaN there (and 0 * Inf = NaN where a throughput channel is 0) — as the reference's f32 film would hold them"""
    kw = dict(amd_mode="wavefront") if org == "wavefront" else dict(amd_mode="fused", amd_deterministic=(org == "fused-deterministic"))
    scene = _bright_cornell(radiance, **kw)
    s_gpu, t_gpu = gpu_render(scene, 64, seed=3)
    s_ref, t_ref, s4, t4, cnt = oracle_render(oracle, scene, 64, seed=3)
    if np.isfinite(radiance).all():
        assert np.nanmax(t_ref) > 2 ** 21
    else:
        assert not np.isfinite(t_ref).all()
    assert_cells_equal(t_gpu, t_ref)
    assert_cells_equal(s_gpu, s_ref, rtol=1e-4)
    got = scene.integrator().last_counters
    for k in ("paths", "rays_closest", "rays_shadow", "splats_issued", "bounces"):
        assert got[k] == cnt[k], k


def test_deterministic_rows_stay_deterministic_below_the_guard(oracle):
    """the guard does not touch ordinary renders: two amd_deterministic renders are still bit for bit equal, and equal to the
    same render with the guard's depth cap exercised (max_depth beyond 64: contributions of depth >= 64 take the f32 ring)"""
    a = make_cornell(width=16, height=12, bins=128, start=3.0, window=30.0, amd_mode="fused", amd_deterministic=True, max_depth=100, rr_depth=90)
    s0, t0 = gpu_render(a, 32)
    s1, t1 = gpu_render(a, 32)
    s_ref, t_ref, s4, t4, cnt = oracle_render(oracle, a, 32)
    assert rel_l2(t0, t_ref) <= 1e-5 and rel_l2(t1, t_ref) <= 1e-5 and rel_l2(s0, s_ref) <= 1e-5
    assert np.array_equal(t0 != 0, t_ref != 0)
    assert a.integrator().last_counters["splats_issued"] == cnt["splats_issued"]
