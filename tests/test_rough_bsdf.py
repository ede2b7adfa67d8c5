"""Microfacet lobes (MTR_BSDF_ROUGHCONDUCTOR / MTR_BSDF_ROUGHPLASTIC: mitsuba's `roughconductor`, `roughplastic` with
distribution = ggx | beckmann, sample_visible = true): the oracle's restatement checked against what a BSDF must satisfy (densities
integrate to one, samples follow the density, weight * pdf = value, reciprocity, energy), the product's arithmetic bit for
bit against the oracle, and renders with these materials through both."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from conftest import hh_render

FP = C.POINTER(C.c_float)


def _mat(kind, alpha=0.1, twosided=False, nonlinear=True, diffuse=(0.5, 0.3, 0.2)):
    """kind: "roughconductor" | "roughplastic", GGX; with "-beckmann" mitsuba's default distribution (no key at all); with
    "-aniso" (roughconductor) alpha_u != alpha_v"""
    import mitransient_amd.mi as mi
    from mitransient_amd.scene import _SceneBuilder
    mi.set_variant("llvm_ad_rgb")
    dist = {} if "-beckmann" in kind else {"distribution": "ggx"}
    if kind.startswith("roughconductor"):
        # "-aniso": alpha_u = alpha, alpha_v = 3 alpha (the lobe is three times wider along the bitangent)
        rough = {"alpha_u": alpha, "alpha_v": 3.0 * alpha} if kind.endswith("-aniso") else {"alpha": alpha}
        bd = {"type": "roughconductor", **dist, **rough, "eta": [1.657, 0.880, 0.521],
              "k": [9.224, 6.270, 4.837], "specular_reflectance": {"type": "rgb", "value": [0.9, 0.8, 0.7]}}
    else:
        bd = {"type": "roughplastic", **dist, "alpha": alpha, "int_ior": 1.5, "ext_ior": 1.0,
              "nonlinear": nonlinear, "diffuse_reflectance": {"type": "rgb", "value": list(diffuse)}}
    if kind.startswith("roughdielectric"):
        rough = {"alpha_u": alpha, "alpha_v": 3.0 * alpha} if kind.endswith("-aniso") else {"alpha": alpha}
        bd = {"type": "roughdielectric", **dist, **rough, "int_ior": 1.5, "ext_ior": 1.0,
              "specular_reflectance": {"type": "rgb", "value": [0.9, 0.8, 0.7]},
              "specular_transmittance": {"type": "rgb", "value": [0.7, 0.9, 0.8]}}
    if kind == "plastic":
        bd = {"type": "plastic", "int_ior": 1.5, "ext_ior": 1.0, "nonlinear": nonlinear,
              "diffuse_reflectance": {"type": "rgb", "value": list(diffuse)}, "specular_reflectance": {"type": "rgb", "value": [0.9, 1.0, 0.8]}}
    if kind == "thindielectric":
        bd = {"type": "thindielectric", "int_ior": 1.5, "ext_ior": 1.0, "specular_reflectance": {"type": "rgb", "value": [0.9, 0.8, 0.7]},
              "specular_transmittance": {"type": "rgb", "value": [0.7, 0.9, 0.8]}}
    if twosided:
        bd = {"type": "twosided", "bsdf": bd}
    return _SceneBuilder({}, ".")._make_material(bd)


def _dirs(n, rng, upper=True):
    v = rng.normal(size=(n, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    if upper:
        v[:, 2] = np.abs(v[:, 2])
    return np.ascontiguousarray(v, np.float32)


def _eval(lib, prefix, m, wi, wo):
    n = len(wi)
    val = np.zeros((n, 3), np.float32); pdf = np.zeros(n, np.float32)
    getattr(lib, prefix + "bsdf_eval_pdf")(C.byref(m), n, wi.ctypes.data_as(FP), wo.ctypes.data_as(FP), val.ctypes.data_as(FP), pdf.ctypes.data_as(FP))
    return val, pdf


def _sample(lib, prefix, m, wi, u1, ua, ub):
    n = len(wi)
    wo = np.zeros((n, 3), np.float32); pdf = np.zeros(n, np.float32); w = np.zeros((n, 3), np.float32)
    getattr(lib, prefix + "bsdf_sample")(C.byref(m), n, wi.ctypes.data_as(FP), u1.ctypes.data_as(FP), ua.ctypes.data_as(FP),
                                         ub.ctypes.data_as(FP), wo.ctypes.data_as(FP), pdf.ctypes.data_as(FP), w.ctypes.data_as(FP))
    return wo, pdf, w


KINDS = ["roughconductor", "roughplastic", "roughconductor-beckmann", "roughplastic-beckmann", "roughconductor-aniso",
         "roughconductor-beckmann-aniso"]


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("alpha", [0.05, 0.1, 0.4])
def test_product_arithmetic_equals_oracle(oracle, host_harness, kind, alpha):
    """every bit of eval / pdf / sample, 2 x 10^5 random direction pairs incl. grazing ones and the two-sided flip"""
    rng = np.random.default_rng(3)
    for twosided in (False, True):
        m = _mat(kind, alpha, twosided)
        n = 200000
        wi, wo = _dirs(n, rng, upper=not twosided), _dirs(n, rng, upper=False)
        wi[:1000, 2] *= 1e-3; wo[1000:2000, 2] *= 1e-3
        wi[2000:2100] = [0, 0, 1]                                          # perpendicular incidence
        v0, p0 = _eval(oracle.lib(), "orc_", m, wi, wo)
        v1, p1 = _eval(host_harness, "hh_", m, wi, wo)
        assert np.array_equal(v0.view(np.uint32), v1.view(np.uint32)) and np.array_equal(p0.view(np.uint32), p1.view(np.uint32))
        assert (p0 > 0).mean() > (0.3 if "beckmann" not in kind else 0.02)      # (Beckmann's tails underflow: D = 0 a few alpha off the peak)
        u = rng.random((3, n)).astype(np.float32)
        a = _sample(oracle.lib(), "orc_", m, wi, u[0], u[1], u[2])
        b = _sample(host_harness, "hh_", m, wi, u[0], u[1], u[2])
        for x, y in zip(a, b):
            assert np.array_equal(x.view(np.uint32), y.view(np.uint32))


def test_special_functions(oracle, host_harness):
    """exp / log / erf / erfinv as the Beckmann lobes use them: oracle and product restate them with the same operation order
    (bit for bit), and both stay within a few ulp of scipy's float64 values"""
    from scipy import special
    rng = np.random.default_rng(0)
    n = 400000
    cases = {0: (rng.uniform(-87, 88, n), np.exp, 2e-7, "rel"),
             1: (np.exp(rng.uniform(-80, 80, n)), np.log, 2e-7, "rel1"),
             2: (rng.uniform(-6, 6, n), special.erf, 3e-6, "abs"),
             3: (np.concatenate([rng.uniform(-1, 1, n // 2), 1 - np.exp(rng.uniform(-16, -2, n // 2))]), special.erfinv, 6e-7, "rel")}
    for which, (x, ref_fn, tol, kind) in cases.items():
        x = np.ascontiguousarray(x, np.float32)
        if which == 3:
            x = np.ascontiguousarray(x[np.abs(x) < 0.9999999])
        a, b = np.zeros_like(x), np.zeros_like(x)
        oracle.lib().orc_special(which, C.c_uint64(len(x)), x.ctypes.data_as(FP), a.ctypes.data_as(FP))
        host_harness.hh_special(which, C.c_uint64(len(x)), x.ctypes.data_as(FP), b.ctypes.data_as(FP))
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), which
        ref = ref_fn(x.astype(np.float64))
        err = np.abs(a - ref)
        if kind == "rel":
            err = err / np.maximum(np.abs(ref), 1e-30)
        elif kind == "rel1":
            err = err / np.maximum(np.abs(ref), 1e-3)
        assert err.max() < tol, (which, err.max())
    # the ends the sampling routine reaches: erf(inf) = 1 (perpendicular incidence), exp(-inf) = 0
    x = np.array([np.inf, -np.inf, 0.0], np.float32); y = np.zeros(3, np.float32)
    oracle.lib().orc_special(2, C.c_uint64(3), x.ctypes.data_as(FP), y.ctypes.data_as(FP))
    assert y.tolist() == [1.0, -1.0, 0.0]
    oracle.lib().orc_special(0, C.c_uint64(3), x.ctypes.data_as(FP), y.ctypes.data_as(FP))
    assert y[1] == 0.0 and y[2] == 1.0


def test_beckmann_is_the_default_distribution_and_differs_from_ggx(oracle):
    """no `distribution` key = mitsuba's default, Beckmann (MTR_MAT_BECKMANN); at equal alpha its lobe has a sharper peak and
    far lighter tails than GGX's: D(0) is the same 1 / (pi alpha^2), the density 3 alpha off the peak is far smaller"""
    from mitransient_amd import _cabi
    mb, mg = _mat("roughconductor-beckmann", 0.1), _mat("roughconductor", 0.1)
    assert mb.flags & _cabi.MTR_MAT_BECKMANN and not (mg.flags & _cabi.MTR_MAT_BECKMANN)
    with pytest.raises(ValueError):
        import mitransient_amd.mi as mi
        from mitransient_amd.scene import _SceneBuilder
        _SceneBuilder({}, ".")._make_material({"type": "roughconductor", "distribution": "phong"})
    wi = np.array([[0.0, 0.0, 1.0]], np.float32)
    for tilt, lo, hi in ((0.0, 0.99, 1.01), (0.3, 0.0, 0.05)):          # half-vector tilted by `tilt` rad: wo = reflection of wi
        wo = np.array([[np.sin(2 * tilt), 0.0, np.cos(2 * tilt)]], np.float32)
        vb, _ = _eval(oracle.lib(), "orc_", mb, wi, wo)
        vg, _ = _eval(oracle.lib(), "orc_", mg, wi, wo)
        assert lo <= vb[0, 0] / vg[0, 0] <= hi, (tilt, vb, vg)


@pytest.mark.parametrize("kind", ["roughconductor-aniso", "roughconductor-beckmann-aniso"])
def test_anisotropic_lobe_is_wider_along_the_bitangent(oracle, kind):
    """alpha_u = 0.1 along the tangent (x), alpha_v = 0.3 along the bitangent (y): the sampled normals spread three times as
    far in y as in x; with alpha_u = alpha_v the flag is not even set and the material is the isotropic one"""
    from mitransient_amd import _cabi
    from mitransient_amd.scene import _SceneBuilder
    m = _mat(kind, 0.1)
    assert m.flags & _cabi.MTR_MAT_ANISOTROPIC and abs(m.c2[0] - 0.3) < 1e-6 and abs(m.alpha - 0.1) < 1e-7
    iso = _SceneBuilder({}, ".")._make_material({"type": "roughconductor", "distribution": "ggx", "alpha_u": 0.2, "alpha_v": 0.2})
    assert not (iso.flags & _cabi.MTR_MAT_ANISOTROPIC) and abs(iso.alpha - 0.2) < 1e-7
    for bad in ({"alpha": 0.1, "alpha_u": 0.1, "alpha_v": 0.2}, {"alpha_u": 0.1}):
        with pytest.raises(ValueError):
            _SceneBuilder({}, ".")._make_material({"type": "roughconductor", **bad})
    with pytest.raises(ValueError):
        _SceneBuilder({}, ".")._make_material({"type": "roughplastic", "alpha_u": 0.1, "alpha_v": 0.2})
    rng = np.random.default_rng(2)
    n = 200000
    wi = np.tile(np.array([[0.0, 0.0, 1.0]], np.float32), (n, 1))
    u = rng.random((3, n)).astype(np.float32)
    wo, pdf, w = _sample(oracle.lib(), "orc_", m, wi, u[0], u[1], u[2])
    ok = pdf > 0
    h = wo[ok] + wi[ok]
    h /= np.linalg.norm(h, axis=1, keepdims=True)
    sx, sy = np.median(np.abs(h[:, 0] / h[:, 2])), np.median(np.abs(h[:, 1] / h[:, 2]))
    assert 2.7 < sy / sx < 3.3, (sx, sy)


def _hemisphere_grid(n_t=512, n_p=1024):
    """midpoint rule in (cos theta, phi): directions and solid-angle weights"""
    ct = (np.arange(n_t) + 0.5) / n_t
    ph = (np.arange(n_p) + 0.5) / n_p * 2 * np.pi
    CT, PH = np.meshgrid(ct, ph, indexing="ij")
    st = np.sqrt(1 - CT * CT)
    d = np.stack([st * np.cos(PH), st * np.sin(PH), CT], -1).reshape(-1, 3).astype(np.float32)
    return np.ascontiguousarray(d), 2 * np.pi / (n_t * n_p)


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("alpha", [0.1, 0.4])
def test_density_integrates_to_one_and_energy_is_bounded(oracle, kind, alpha):
    m = _mat(kind, alpha)
    wo, dw = _hemisphere_grid()
    for mu in (0.95, 0.6, 0.25):
        wi = np.tile(np.array([[np.sqrt(1 - mu * mu), 0, mu]], np.float32), (len(wo), 1))
        val, pdf = _eval(oracle.lib(), "orc_", m, wi, wo)
        total = pdf.astype(np.float64).sum() * dw
        # the density integrates to the probability that a sample is usable at all: a visible normal can reflect wi below
        # the horizon (more often at grazing incidence and large alpha), and that part of the lobe is lost
        rng = np.random.default_rng(1)
        ns = 200000
        u = rng.random((3, ns)).astype(np.float32)
        swo, spdf, sw = _sample(oracle.lib(), "orc_", m, np.ascontiguousarray(wi[:ns]), u[0], u[1], u[2])
        usable = ((swo[:, 2] > 0) & (spdf > 0)).mean()
        assert abs(total - usable) < 0.01 and 0.5 < total <= 1.001, (kind, alpha, mu, total, usable)
        if mu > 0.9 and alpha <= 0.1 and not kind.endswith("-aniso"):       # (alpha_v = 3 alpha loses more of the lobe below the horizon)
            assert total > 0.98
        albedo = val.astype(np.float64).sum(0) * dw
        assert np.all(albedo <= 1.0 + 1e-3) and np.all(albedo > 0.05)


@pytest.mark.parametrize("kind", KINDS)
def test_samples_follow_the_density_and_weights_are_value_over_pdf(oracle, kind):
    m = _mat(kind, 0.25)
    rng = np.random.default_rng(5)
    n = 400000
    mu = 0.7
    wi = np.tile(np.array([[np.sqrt(1 - mu * mu), 0, mu]], np.float32), (n, 1))
    u = rng.random((3, n)).astype(np.float32)
    wo, pdf, w = _sample(oracle.lib(), "orc_", m, wi, u[0], u[1], u[2])
    ok = (pdf > 0) & (w.max(1) > 0)
    assert ok.mean() > (0.9 if not kind.endswith("-aniso") else 0.7)          # (alpha_v = 0.75: a quarter of the visible normals reflect below the horizon)
    val, pdf2 = _eval(oracle.lib(), "orc_", m, wi[ok], wo[ok])
    # the density reported with the sample is the density of that direction, and weight * pdf is the value
    assert np.allclose(pdf[ok], pdf2, rtol=2e-4, atol=1e-6)
    assert np.allclose(w[ok] * pdf[ok, None], val, rtol=2e-3, atol=1e-6)
    # histogram of the sampled directions against the integrated density
    n_t, n_p = 16, 16
    it = np.minimum((wo[ok, 2] * n_t).astype(int), n_t - 1)
    ip = np.minimum(((np.arctan2(wo[ok, 1], wo[ok, 0]) / (2 * np.pi)) % 1.0 * n_p).astype(int), n_p - 1)
    hist = np.bincount(it * n_p + ip, minlength=n_t * n_p).astype(np.float64) / n
    g, dw = _hemisphere_grid(256, 512)
    _, pg = _eval(oracle.lib(), "orc_", m, np.tile(wi[:1], (len(g), 1)), g)
    gt = np.minimum((g[:, 2] * n_t).astype(int), n_t - 1)
    gp = np.minimum(((np.arctan2(g[:, 1], g[:, 0]) / (2 * np.pi)) % 1.0 * n_p).astype(int), n_p - 1)
    expect = np.bincount(gt * n_p + gp, weights=pg.astype(np.float64) * dw, minlength=n_t * n_p)
    big = expect > 2e-3
    assert big.sum() > 20
    sigma = np.sqrt(expect[big] / n)
    assert np.all(np.abs(hist[big] - expect[big]) < 6 * sigma + 0.02 * expect[big])     # 2 %: the midpoint rule across cell borders


@pytest.mark.parametrize("kind", KINDS)
def test_reciprocity(oracle, kind):
    """f(wi, wo) = value / cos(theta_o) is symmetric"""
    m = _mat(kind, 0.3)
    rng = np.random.default_rng(9)
    a, b = _dirs(20000, rng), _dirs(20000, rng)
    keep = (a[:, 2] > 0.05) & (b[:, 2] > 0.05)
    a, b = np.ascontiguousarray(a[keep]), np.ascontiguousarray(b[keep])
    v_ab, _ = _eval(oracle.lib(), "orc_", m, a, b)
    v_ba, _ = _eval(oracle.lib(), "orc_", m, b, a)
    assert np.allclose(v_ab / b[:, 2:3], v_ba / a[:, 2:3], rtol=2e-3, atol=1e-6)


def test_rough_plastic_tables():
    from mitransient_amd.microfacet import rough_plastic_tables
    ext, internal = rough_plastic_tables(0.001, 1.5)
    assert abs(ext[-1] - 0.96) < 1e-3                      # smooth limit at normal incidence: 1 - ((n - 1) / (n + 1))^2
    assert 0.55 < internal < 0.62                          # ~ the diffuse Fresnel reflectance from inside (0.596 for n = 1.5)
    ext, internal = rough_plastic_tables(0.1, 1.5)
    assert np.all(np.diff(ext[8:]) > -1e-3) and 0.9 < ext[-1] < 0.97 and 0.3 < ext[0] < 0.8


# ---- roughdielectric: a transmissive lobe — directions on the whole sphere, wi from either side ---------------------------
DKINDS = ["roughdielectric", "roughdielectric-beckmann", "roughdielectric-aniso"]


def _sphere_grid(n_t=512, n_p=1024):
    ct = (np.arange(n_t) + 0.5) / n_t * 2 - 1
    ph = (np.arange(n_p) + 0.5) / n_p * 2 * np.pi
    CT, PH = np.meshgrid(ct, ph, indexing="ij")
    st = np.sqrt(1 - CT * CT)
    d = np.stack([st * np.cos(PH), st * np.sin(PH), CT], -1).reshape(-1, 3).astype(np.float32)
    return np.ascontiguousarray(d), 4 * np.pi / (n_t * n_p)


@pytest.mark.parametrize("kind", DKINDS)
@pytest.mark.parametrize("alpha", [0.05, 0.3])
def test_rough_dielectric_product_equals_oracle(oracle, host_harness, kind, alpha):
    """every bit of eval / pdf / sample for directions on the whole sphere (wi inside and outside, grazing, perpendicular)"""
    rng = np.random.default_rng(4)
    m = _mat(kind, alpha)
    n = 200000
    wi, wo = _dirs(n, rng, upper=False), _dirs(n, rng, upper=False)
    wi[:1000, 2] *= 1e-3; wo[1000:2000, 2] *= 1e-3
    wi[2000:2050] = [0, 0, 1]; wi[2050:2100] = [0, 0, -1]
    v0, p0 = _eval(oracle.lib(), "orc_", m, wi, wo)
    v1, p1 = _eval(host_harness, "hh_", m, wi, wo)
    assert np.array_equal(v0.view(np.uint32), v1.view(np.uint32)) and np.array_equal(p0.view(np.uint32), p1.view(np.uint32))
    assert (p0 > 0).mean() > 0.02 and np.isfinite(v0).all() and np.isfinite(p0).all()
    u = rng.random((3, n)).astype(np.float32)
    a = _sample(oracle.lib(), "orc_", m, wi, u[0], u[1], u[2])
    b = _sample(host_harness, "hh_", m, wi, u[0], u[1], u[2])
    for x, y in zip(a, b):
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
    assert np.isfinite(a[0]).all() and np.isfinite(a[1]).all() and np.isfinite(a[2]).all()


@pytest.mark.parametrize("kind", DKINDS)
@pytest.mark.parametrize("side", [1.0, -1.0], ids=["from-outside", "from-inside"])
def test_rough_dielectric_density_samples_and_weights(oracle, kind, side):
    """the density integrates over the SPHERE to the usable-sample fraction; the density reported with a sample is the density
    of that direction; weight x pdf = value; the sampled directions follow the density (both lobes, either side); reflected
    plus transmitted energy stays below one"""
    m = _mat(kind, 0.25)
    g, dw = _sphere_grid(1536, 3072)             # (fine: a Beckmann lobe falls off fast across a histogram cell)
    rng = np.random.default_rng(6)
    for mu in (0.9, 0.5):
        wi1 = np.array([[np.sqrt(1 - mu * mu), 0, side * mu]], np.float32)
        val, pdf = _eval(oracle.lib(), "orc_", m, np.tile(wi1, (len(g), 1)), g)
        total = pdf.astype(np.float64).sum() * dw
        n = 400000
        wi = np.tile(wi1, (n, 1))
        u = rng.random((3, n)).astype(np.float32)
        wo, spdf, w = _sample(oracle.lib(), "orc_", m, wi, u[0], u[1], u[2])
        ok = (spdf > 0) & (w.max(1) > 0)
        assert abs(total - ok.mean()) < 0.015 and 0.5 < total <= 1.002, (kind, side, mu, total, ok.mean())
        sval, spdf2 = _eval(oracle.lib(), "orc_", m, np.ascontiguousarray(wi[ok]), np.ascontiguousarray(wo[ok]))
        assert np.allclose(spdf[ok], spdf2, rtol=5e-4, atol=1e-6)
        assert np.allclose(w[ok] * spdf[ok, None], sval, rtol=3e-3, atol=1e-6)
        # both lobes are taken, in the proportion the densities of the two hemispheres integrate to
        up = g[:, 2] * side > 0
        frac_r = pdf[up].astype(np.float64).sum() * dw / total
        assert abs((wo[ok, 2] * side > 0).mean() - frac_r) < 0.01 and 0.01 < frac_r < 0.995          # (from inside at 60 degrees nearly everything is reflected: past the critical angle)
        # histogram on the sphere
        n_t, n_p = 24, 16
        cell = lambda d: np.minimum(((d[:, 2] + 1) / 2 * n_t).astype(int), n_t - 1) * n_p + \
            np.minimum(((np.arctan2(d[:, 1], d[:, 0]) / (2 * np.pi)) % 1.0 * n_p).astype(int), n_p - 1)
        hist = np.bincount(cell(wo[ok]), minlength=n_t * n_p).astype(np.float64) / n
        expect = np.bincount(cell(g), weights=pdf.astype(np.float64) * dw, minlength=n_t * n_p)
        big = expect > 5e-4
        assert big.sum() > 5                         # (Beckmann's lobes are narrow: a handful of cells hold them)
        assert np.all(np.abs(hist[big] - expect[big]) < 6 * np.sqrt(expect[big] / n) + 0.03 * expect[big])
        # energy: reflectance + transmittance (radiance leaving into a medium of index n carries n^2: undo the 1 / eta^2 scale)
        eta = 1.5 if side > 0 else 1 / 1.5
        refl = val[up].astype(np.float64).sum(0) * dw
        tran = val[~up].astype(np.float64).sum(0) * dw * eta * eta
        assert np.all(refl / np.array([0.9, 0.8, 0.7]) + tran / np.array([0.7, 0.9, 0.8]) <= 1.0 + 5e-3)
        assert np.all(refl + tran > 0.3)


def test_plastic_and_thindielectric(oracle, host_harness):
    """`plastic` (a delta coat over a diffuse base: eval / pdf see the base, the lobes are chosen by F_i and the sampling weight)
    and `thindielectric` (two delta lobes, r' = 2r / (1 + r), straight through): product = oracle bit for bit, densities, weights
    and energy as the plugins define them"""
    rng = np.random.default_rng(8)
    n = 200000
    for kind, twosided in (("plastic", False), ("plastic", True), ("thindielectric", False)):
        m = _mat(kind, twosided=twosided)
        wi, wo = _dirs(n, rng, upper=(kind == "plastic" and not twosided)), _dirs(n, rng, upper=False)
        wi[:500, 2] *= 1e-3
        v0, p0 = _eval(oracle.lib(), "orc_", m, wi, wo)
        v1, p1 = _eval(host_harness, "hh_", m, wi, wo)
        assert np.array_equal(v0.view(np.uint32), v1.view(np.uint32)) and np.array_equal(p0.view(np.uint32), p1.view(np.uint32))
        u = rng.random((3, n)).astype(np.float32)
        a = _sample(oracle.lib(), "orc_", m, wi, u[0], u[1], u[2])
        b = _sample(host_harness, "hh_", m, wi, u[0], u[1], u[2])
        for x, y in zip(a, b):
            assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
    # plastic at one incidence: the base's density integrates to the probability of taking the base
    m = _mat("plastic")
    g, dw = _hemisphere_grid()
    mu = 0.6
    wi1 = np.array([[np.sqrt(1 - mu * mu), 0, mu]], np.float32)
    val, pdf = _eval(oracle.lib(), "orc_", m, np.tile(wi1, (len(g), 1)), g)
    p_base = pdf.astype(np.float64).sum() * dw
    n = 300000
    wi = np.tile(wi1, (n, 1))
    u = rng.random((3, n)).astype(np.float32)
    wo, spdf, w = _sample(oracle.lib(), "orc_", m, wi, u[0], u[1], u[2])
    mirror = np.all(np.abs(wo - wi * np.array([-1, -1, 1], np.float32)) < 1e-6, axis=1)
    assert abs(mirror.mean() - (1 - p_base)) < 0.005 and 0.02 < mirror.mean() < 0.5
    sval, spdf2 = _eval(oracle.lib(), "orc_", m, np.ascontiguousarray(wi[~mirror]), np.ascontiguousarray(wo[~mirror]))
    assert np.allclose(spdf[~mirror], spdf2, rtol=2e-4) and np.allclose(w[~mirror] * spdf[~mirror, None], sval, rtol=2e-3, atol=1e-7)
    # the coat: weight x probability = specular_reflectance x F(cos 0.6, 1.5); Fresnel by the formula
    ci, eta = mu, 1.5
    ct = np.sqrt(1 - (1 - ci * ci) / eta ** 2)
    F = 0.5 * (((ci - eta * ct) / (ci + eta * ct)) ** 2 + ((ct - eta * ci) / (ct + eta * ci)) ** 2)
    assert np.allclose(w[mirror][0] * spdf[mirror][0], np.array([0.9, 1.0, 0.8]) * F, rtol=1e-4)
    albedo = val.astype(np.float64).sum(0) * dw + np.array([0.9, 1.0, 0.8]) * F
    assert np.all(albedo <= 1.0) and np.all(albedo > 0.1)
    # thin slab: reflect with r' = 2r / (1 + r), else straight through; from either side; eta stays 1
    m = _mat("thindielectric")
    wi = np.tile(np.array([[0.6, 0.0, -0.8]], np.float32), (n, 1))
    wo, spdf, w = _sample(oracle.lib(), "orc_", m, wi, u[0], u[1], u[2])
    ci = 0.8; ct = np.sqrt(1 - (1 - ci * ci) / eta ** 2)
    r = 0.5 * (((ci - eta * ct) / (ci + eta * ct)) ** 2 + ((ct - eta * ci) / (ct + eta * ci)) ** 2)
    r2 = 2 * r / (1 + r)
    refl = wo[:, 2] < 0
    assert abs(refl.mean() - r2) < 0.004
    assert np.allclose(wo[refl], [-0.6, 0.0, -0.8]) and np.allclose(wo[~refl], [-0.6, 0.0, 0.8])
    assert np.allclose(w[refl], [0.9, 0.8, 0.7]) and np.allclose(w[~refl], [0.7, 0.9, 0.8])
    assert np.allclose(spdf[refl], r2, rtol=1e-5) and np.allclose(spdf[~refl], 1 - r2, rtol=1e-5)


def test_rough_dielectric_plugin_rules():
    from mitransient_amd import _cabi
    from mitransient_amd.scene import _SceneBuilder
    mk = lambda bd: _SceneBuilder({}, ".")._make_material(bd)
    m = mk({"type": "roughdielectric"})                                   # mitsuba's defaults: beckmann, alpha 0.1, bk7 in air
    assert m.type == _cabi.MTR_BSDF_ROUGHDIELECTRIC and m.flags & _cabi.MTR_MAT_BECKMANN and abs(m.alpha - 0.1) < 1e-7
    assert abs(m.int_ior - 1.5046) < 1e-4 and abs(m.ext_ior - 1.000277) < 1e-5
    m = mk({"type": "roughdielectric", "distribution": "ggx", "alpha_u": 0.1, "alpha_v": 0.2})
    assert m.flags & _cabi.MTR_MAT_ANISOTROPIC and abs(m.b[0] - 0.2) < 1e-7 and m.c2[0] == 1.0
    with pytest.raises(ValueError):
        mk({"type": "twosided", "bsdf": {"type": "roughdielectric"}})     # transmissive: not under twosided
    with pytest.raises(ValueError):
        mk({"type": "roughdielectric", "int_ior": 1.3, "ext_ior": 1.3})


def test_rough_dielectric_tends_to_the_smooth_dielectric(oracle):
    """alpha -> 0: a glass box with a barely rough interface renders (steady image, mean over the frame) like the `dielectric`
    one — reflection, refraction, the eta bookkeeping of paths inside and the radiance scaling all enter"""
    import mitransient_amd as mitr
    import mitransient_amd.mi as mi
    mi.set_variant("llvm_ad_rgb")
    means = {}
    for name, bsdf in (("rough", {"type": "roughdielectric", "distribution": "ggx", "alpha": 0.002, "int_ior": 1.5, "ext_ior": 1.0}),
                       ("smooth", {"type": "dielectric", "int_ior": 1.5, "ext_ior": 1.0})):
        d = mitr.cornell_box()
        d["sensor"]["film"].update(width=16, height=16, temporal_bins=8, start_opl=0.0, bin_width_opl=4.0)
        d["integrator"].update(max_depth=12, rr_depth=20)
        d["small-box"]["bsdf"] = bsdf
        d["large-box"]["bsdf"] = bsdf
        scene = mi.load_dict(d)
        p = scene.integrator().render_params(scene.sensors()[0].film(), 0, 256)
        t4, s4, cnt = oracle.render(scene.data(), p)
        img = s4[..., :3] / s4[..., 3:4]
        means[name] = (img.mean(), img.std() / np.sqrt(img.size), cnt["rays_closest"])
    (a, sa, ra), (b, sb, rb) = means["rough"], means["smooth"]
    assert abs(a - b) < 0.03 * b + 6 * (sa + sb), (means)
    assert abs(ra - rb) < 0.02 * rb                       # paths are as long: the same number of rays


def _rough_cornell(distribution="ggx", **film):
    """distribution: "ggx", "beckmann", or None = no key at all (mitsuba's default: Beckmann)"""
    import mitransient_amd as mitr
    import mitransient_amd.mi as mi
    mi.set_variant("llvm_ad_rgb")
    aniso = distribution == "aniso"            # Beckmann by default + anisotropic roughconductors (one of them GGX)
    dk = {} if distribution in (None, "glass", "plastic") or aniso else {"distribution": distribution}
    d = mitr.cornell_box()
    d["sensor"]["film"].update(width=24, height=24, temporal_bins=64, start_opl=3.5, bin_width_opl=6.0 / 64)
    d["sensor"]["film"].update(film)
    d["floor"]["bsdf"] = {"type": "roughplastic", **dk, "alpha": 0.1, "int_ior": 1.5, "nonlinear": True,
                          "diffuse_reflectance": {"type": "rgb", "value": [0.58, 0.42, 0.3]}}
    d["large-box"]["bsdf"] = {"type": "roughconductor", **dk, "alpha": 0.15, "eta": [1.657, 0.880, 0.521],
                              "k": [9.224, 6.270, 4.837]}
    d["back"]["bsdf"] = {"type": "twosided", "bsdf": {"type": "roughplastic", **dk, "alpha": 0.3,
                                                      "diffuse_reflectance": {"type": "rgb", "value": [0.2, 0.5, 0.7]}}}
    d["small-box"]["bsdf"] = {"type": "twosided", "bsdf": {"type": "roughconductor", **dk, "alpha": 0.05,
                                                           "eta": 0.2, "k": 3.9}}
    if distribution == "glass":                # rough refractive boxes: paths go through them (eta changes, both lobes, both sides)
        d["small-box"]["bsdf"] = {"type": "roughdielectric", "distribution": "ggx", "alpha": 0.1, "int_ior": 1.5, "ext_ior": 1.0}
        d["large-box"]["bsdf"] = {"type": "roughdielectric", "alpha_u": 0.05, "alpha_v": 0.2, "int_ior": "water",
                                  "specular_transmittance": {"type": "rgb", "value": [0.8, 0.95, 0.9]}}
    if distribution == "plastic":              # a plastic floor (nonlinear), a thin glass pane as the small box, a two-sided plastic back wall
        d["floor"]["bsdf"] = {"type": "plastic", "int_ior": 1.5, "nonlinear": True, "diffuse_reflectance": {"type": "rgb", "value": [0.58, 0.42, 0.3]}}
        d["back"]["bsdf"] = {"type": "twosided", "bsdf": {"type": "plastic", "diffuse_reflectance": {"type": "rgb", "value": [0.2, 0.5, 0.7]},
                                                          "specular_reflectance": {"type": "rgb", "value": [0.8, 0.8, 0.8]}}}
        d["small-box"]["bsdf"] = {"type": "thindielectric", "specular_transmittance": {"type": "rgb", "value": [0.9, 0.95, 0.85]}}
    if aniso:
        d["large-box"]["bsdf"].pop("alpha"); d["large-box"]["bsdf"].update(alpha_u=0.1, alpha_v=0.3)
        d["small-box"]["bsdf"]["bsdf"].pop("alpha"); d["small-box"]["bsdf"]["bsdf"].update(alpha_u=0.15, alpha_v=0.05, distribution="ggx")
    return d


@pytest.mark.parametrize("distribution", ["ggx", None, "aniso", "glass", "plastic"],
                         ids=["ggx", "beckmann-by-default", "anisotropic", "roughdielectric", "plastic-thindielectric"])
@pytest.mark.parametrize("wide", [0, 1], ids=["bvh2", "wide-8"])
def test_host_harness_rough_scene_bit_for_bit(oracle, host_harness, wide, distribution):
    import mitransient_amd.mi as mi
    d = _rough_cornell(distribution)
    d["integrator"].update(max_depth=-1, rr_depth=4)
    scene = mi.load_dict(d)
    sd = scene.data()
    p = scene.integrator().render_params(scene.sensors()[0].film(), 3, 24)
    t4, s4, cnt = oracle.render(sd, p, n_threads=1)
    host_harness.hh_set_node_pairs(wide); host_harness.hh_set_wide(wide)
    try:
        ht, hs, hc = hh_render(host_harness, sd, p)
    finally:
        host_harness.hh_set_node_pairs(0); host_harness.hh_set_wide(0)
    assert np.array_equal(t4, ht) and np.array_equal(s4, hs)
    for k in ("paths", "rays_closest", "rays_shadow", "splats_issued", "bounces"):
        assert hc[k] == cnt[k]
    assert np.count_nonzero(t4) > 3000 and np.isfinite(t4).all()


@pytest.mark.parametrize("distribution", ["ggx", "beckmann", "glass", "plastic"])
def test_energy_identity_with_rough_materials(oracle, distribution):
    """transient.sum(time) == steady when the window holds every path (1-simple-nlos-scenes.ipynb md cell 8)"""
    import mitransient_amd.mi as mi
    d = _rough_cornell(distribution, start_opl=0.0, bin_width_opl=1.0, temporal_bins=128)
    scene = mi.load_dict(d)
    p = scene.integrator().render_params(scene.sensors()[0].film(), 0, 32)
    t4, s4, cnt = oracle.render(scene.data(), p)
    steady = s4[..., :3] / np.maximum(s4[..., 3:4], 1)
    assert np.allclose(t4[..., :3].sum(2), steady, rtol=2e-4, atol=1e-6)
    assert steady.mean() > 0.05


@pytest.mark.parametrize("distribution", ["ggx", "beckmann"])
def test_rough_lobes_tend_to_their_smooth_limits(oracle, distribution):
    """alpha -> 0: the steady image of a roughconductor wall tends to the conductor's (same eta / k); compared in the mean
    over the image at a few hundred samples per pixel (k sigma)"""
    import mitransient_amd as mitr
    import mitransient_amd.mi as mi
    mi.set_variant("llvm_ad_rgb")
    means = {}
    for name, bsdf in (("rough", {"type": "roughconductor", "distribution": distribution, "alpha": 0.002, "eta": 0.2, "k": 3.9}),
                       ("smooth", {"type": "conductor", "eta": 0.2, "k": 3.9})):
        d = mitr.cornell_box()
        d["sensor"]["film"].update(width=16, height=16, temporal_bins=8, start_opl=0.0, bin_width_opl=4.0)
        d["back"]["bsdf"] = bsdf
        scene = mi.load_dict(d)
        p = scene.integrator().render_params(scene.sensors()[0].film(), 0, 256)
        t4, s4, _ = oracle.render(scene.data(), p)
        img = s4[..., :3] / s4[..., 3:4]
        means[name] = (img.mean(), img.std() / np.sqrt(img.size))
    (a, sa), (b, sb) = means["rough"], means["smooth"]
    assert abs(a - b) < 0.03 * b + 6 * (sa + sb)
