"""Host-side tables of the rough dielectric coat of ``roughplastic``.

mitsuba computes them when the plugin is built (``RoughPlastic::parameters_changed``): the transmittance of the rough
interface for 64 incidence cosines, ``eval_transmittance(distr, wi, eta)``, and the mean reflectance seen from inside,
``mean(eval_reflectance(distr, wi, 1 / eta) * wi.z) * 2`` — both Gauss-Legendre quadratures (32 x 32 nodes when
eta > 1, 128 x 128 otherwise) of the visible-normal sampling estimator over the unit square.  This module restates
that quadrature in float64 numpy (GGX and Beckmann, isotropic alpha); the library only interpolates the table
(``mtr_material.external_transmittance``).  [upstream: mitsuba3 src/bsdfs/roughplastic.cpp, include/mitsuba/render/microfacet.h;
not present under the reference tree — restated from the published source.]
"""
from __future__ import annotations

import numpy as np

RES = 64     # MI_ROUGH_TRANSMITTANCE_RES


def _disk(u1, u2):
    """warp::square_to_uniform_disk_concentric"""
    x, y = 2.0 * u1 - 1.0, 2.0 * u2 - 1.0
    swap = np.abs(x) < np.abs(y)
    r = np.where(swap, y, x)
    rp = np.where(swap, x, y)
    with np.errstate(divide="ignore", invalid="ignore"):
        phi = 0.25 * np.pi * rp / r
    phi = np.where((x == 0) & (y == 0), 0.0, phi)
    s, c = np.sin(phi), np.cos(phi)
    return r * np.where(swap, s, c), r * np.where(swap, c, s)


def _erf_pair():
    """erf / erfinv in float64: scipy's when it is installed, else numpy restatements (math.erf element-wise; erfinv by
    M. Giles' single-precision polynomial as the first guess and three Newton steps on erf) — scipy is NOT a dependency
    of loading a default (Beckmann) `roughplastic` scene"""
    try:
        from scipy.special import erf, erfinv
        return erf, erfinv
    except ImportError:
        pass
    import math
    erf = np.vectorize(math.erf, otypes=[np.float64])

    def erfinv(x):
        x = np.asarray(x, dtype=np.float64)
        with np.errstate(divide="ignore", invalid="ignore"):
            w = -np.log((1.0 - x) * (1.0 + x))
            w1 = w - 2.5
            p1 = 2.81022636e-08
            for c in (3.43273939e-07, -3.5233877e-06, -4.39150654e-06, 0.00021858087, -0.00125372503, -0.00417768164, 0.246640727, 1.50140941):
                p1 = p1 * w1 + c
            w2 = np.sqrt(np.maximum(w, 5.0)) - 3.0
            p2 = -0.000200214257
            for c in (0.000100950558, 0.00134934322, -0.00367342844, 0.00573950773, -0.0076224613, 0.00943887047, 1.00167406, 2.83297682):
                p2 = p2 * w2 + c
            y = np.where(w < 5.0, p1, p2) * x
            for _ in range(3):
                y = y - (erf(y) - x) * (0.5 * np.sqrt(np.pi)) * np.exp(y * y)
        return np.where(np.abs(x) >= 1.0, np.copysign(np.inf, x), y)
    return erf, erfinv


def _beckmann_slopes(ct, u1, u2):
    """MicrofacetDistribution::sample_visible_11 (Beckmann): inversion of the visible-slope CDF in the erf domain, a
    closed-form first guess + three Newton iterations, as mitsuba does it (float64 here)"""
    erf, erfinv = _erf_pair()
    inv_sqrt_pi = 1.0 / np.sqrt(np.pi)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        tan_t = np.sqrt(np.maximum(1.0 - ct * ct, 0.0)) / ct
        cot_t = 1.0 / tan_t
        maxval = erf(cot_t)
        u1 = np.clip(u1, 1e-6, 1.0 - 1e-6); u2 = np.clip(u2, 1e-6, 1.0 - 1e-6)
        x = maxval - (maxval + 1.0) * erf(np.sqrt(-np.log(u1)))
        tail = np.where(tan_t == 0, 0.0, inv_sqrt_pi * tan_t * np.exp(-cot_t * cot_t))
        target = u1 * (1.0 + maxval + tail)
        for _ in range(3):
            x = np.clip(x, -0.999999, 0.999999)
            slope = erfinv(x)
            value = 1.0 + x + inv_sqrt_pi * tan_t * np.exp(-slope * slope) - target
            x = x - value / (1.0 - slope * tan_t)
        x = np.clip(x, -0.999999, 0.999999)
        return erfinv(x), erfinv(2.0 * u2 - 1.0) + 0.0 * x


def _ggx_sample(wi, alpha, u1, u2, beckmann=False):
    """MicrofacetDistribution::sample (visible normals); wi (..., 3) with wi.z > 0"""
    wp = np.stack([alpha * wi[..., 0], alpha * wi[..., 1], wi[..., 2]], -1)
    wp = wp / np.linalg.norm(wp, axis=-1, keepdims=True)
    s2 = 1.0 - wp[..., 2] ** 2
    flat = np.abs(s2) <= 4.0 * 2.0 ** -24
    inv = 1.0 / np.sqrt(np.where(flat, 1.0, s2))
    sin_phi = np.where(flat, 0.0, np.clip(wp[..., 1] * inv, -1, 1))
    cos_phi = np.where(flat, 1.0, np.clip(wp[..., 0] * inv, -1, 1))
    ct = wp[..., 2]
    if beckmann:
        sx, sy = _beckmann_slopes(ct, np.broadcast_to(u1, ct.shape), np.broadcast_to(u2, ct.shape))
    else:
        px, py = _disk(u1, u2)
        s = 0.5 * (1.0 + ct)
        a = np.sqrt(np.maximum(1.0 - px * px, 0.0))
        py = a + (py - a) * s
        z = np.sqrt(np.maximum(1.0 - px * px - py * py, 0.0))
        st = np.sqrt(np.maximum(1.0 - ct * ct, 0.0))
        norm = 1.0 / (st * py + ct * z)
        sx, sy = (ct * py - st * z) * norm, px * norm
    rx = (cos_phi * sx - sin_phi * sy) * alpha
    ry = (sin_phi * sx + cos_phi * sy) * alpha
    m = np.stack([-rx, -ry, np.ones_like(rx)], -1)
    return m / np.linalg.norm(m, axis=-1, keepdims=True)


def _g1(v, m, alpha, beckmann=False):
    """MicrofacetDistribution::smith_g1"""
    xy = (alpha * v[..., 0]) ** 2 + (alpha * v[..., 1]) ** 2
    with np.errstate(divide="ignore", invalid="ignore"):
        t = xy / v[..., 2] ** 2
        if beckmann:
            a = 1.0 / np.sqrt(t)
            r = np.where(a >= 1.6, 1.0, (3.535 * a + 2.181 * a * a) / (1.0 + 2.276 * a + 2.577 * a * a))
        else:
            r = 2.0 / (1.0 + np.sqrt(1.0 + t))
    r = np.where(xy == 0, 1.0, r)
    return np.where(np.sum(v * m, -1) * v[..., 2] <= 0, 0.0, r)


def _fresnel(cos_i, eta):
    """fresnel(cos_theta_i, eta) -> (r, cos_theta_t, eta_it, eta_ti)"""
    outside = cos_i >= 0
    eta_it = np.where(outside, eta, 1.0 / eta)
    eta_ti = np.where(outside, 1.0 / eta, eta)
    ct2 = 1.0 - (1.0 - cos_i * cos_i) * eta_ti * eta_ti
    ci = np.abs(cos_i)
    ct = np.sqrt(np.maximum(ct2, 0.0))
    with np.errstate(divide="ignore", invalid="ignore"):
        a_s = (ci - eta_it * ct) / (ci + eta_it * ct)
        a_p = (ct - eta_it * ci) / (ct + eta_it * ci)
    r = 0.5 * (a_s * a_s + a_p * a_p)
    if eta == 1.0:
        r = np.zeros_like(r)
    r = np.where(ci == 0, 1.0, r)
    return r, np.where(cos_i >= 0, -ct, ct), eta_it, eta_ti


def _quadrature(eta):
    res = 32 if eta > 1 else 128
    nodes, weights = np.polynomial.legendre.leggauss(res)
    u = 0.5 * nodes + 0.5
    u1, u2 = np.meshgrid(u, u, indexing="ij")
    w = np.outer(weights, weights)
    return u1, u2, w


def eval_reflectance(alpha, mu, eta, beckmann=False):
    """eval_reflectance(distr, wi = (sqrt(1 - mu^2), 0, mu), eta), one value per mu"""
    u1, u2, w = _quadrature(eta)
    mu = np.asarray(mu, np.float64)
    wi = np.stack([np.sqrt(1.0 - mu * mu), np.zeros_like(mu), mu], -1)[:, None, None, :]
    m = _ggx_sample(np.broadcast_to(wi, (len(mu),) + u1.shape + (3,)), alpha, u1[None], u2[None], beckmann)
    d = np.sum(wi * m, -1)
    wo = 2.0 * d[..., None] * m - wi
    f = _fresnel(d, eta)[0]
    smith = _g1(wo, m, alpha, beckmann) * f
    smith = np.where((wo[..., 2] <= 0) | (f <= 0), 0.0, smith)
    return np.sum(smith * w[None], axis=(1, 2)) * 0.25


def eval_transmittance(alpha, mu, eta, beckmann=False):
    """eval_transmittance(distr, wi, eta)"""
    u1, u2, w = _quadrature(eta)
    mu = np.asarray(mu, np.float64)
    wi = np.stack([np.sqrt(1.0 - mu * mu), np.zeros_like(mu), mu], -1)[:, None, None, :]
    m = _ggx_sample(np.broadcast_to(wi, (len(mu),) + u1.shape + (3,)), alpha, u1[None], u2[None], beckmann)
    d = np.sum(wi * m, -1)
    f, cos_t, eta_it, eta_ti = _fresnel(d, eta)
    wo = m * (d * eta_ti + cos_t)[..., None] - wi * eta_ti[..., None]          # refract(wi, m, cos_theta_t, eta_ti)
    smith = _g1(wo, m, alpha, beckmann) * (1.0 - f)
    smith = np.where((wo[..., 2] >= 0) | (f >= 1), 0.0, smith)
    return np.sum(smith * w[None], axis=(1, 2)) * 0.25


_cache = {}


def rough_plastic_tables(alpha: float, eta: float, distribution: str = "ggx"):
    """(external_transmittance float32[64], internal_reflectance float32) of RoughPlastic::parameters_changed"""
    key = (float(np.float32(alpha)), float(np.float32(eta)), distribution)
    if key not in _cache:
        a, e, _ = key
        beck = distribution == "beckmann"
        mu = np.maximum(1e-6, np.linspace(0.0, 1.0, RES))
        ext = eval_transmittance(a, mu, e, beck)
        internal = float(np.mean(eval_reflectance(a, mu, 1.0 / e, beck) * mu) * 2.0)
        _cache[key] = (ext.astype(np.float32), np.float32(internal))
    return _cache[key]
