// mtr_nlos.h — per-path arithmetic of the NLOS tier (PRODUCT code, host+device like mtr_core.h).
//
// transient_nlos_path (mitransient/integrators/transientnlospath.py) + nlos_capture_meter
// (mitransient/sensors/nloscapturemeter.py) + mitsuba's `projector`:
//   nlos_begin()    ADIntegrator.sample_rays + NLOSCaptureMeter.sample_ray (:136-202), loop init (:712-723)
//   nlos_bounce()   one iteration of TransientNLOSPath.sample (:740-918): closest hit, laser sampling
//                   (:511-635) or plain emitter sampling (:432-509), hidden-geometry / BSDF sampling
//                   (:637-670, :797-833), Russian roulette
// The Mitsuba pieces (Projector::sample_direction, Rectangle/Mesh::sample_position,
// DiscreteDistribution::sample_reuse_pmf) follow their published algorithms [upstream-unverified].
#pragma once
#include "mtr_core.h"

namespace mtr {

constexpr float kDrEps = 5.9604644775390625e-8f;      // dr.epsilon(Float) = 2^-24

struct alignas(16) NlosShape {       // 64 B
    float center[3], du[3], dv[3], n[3];
    uint32_t first_tri, n_tris, is_rect;
    float inv_area;
};

struct NlosConst {
    f3 sensor_origin;
    f3 w_center, w_du, w_dv;          // relay-wall rectangle
    f3 l_origin, l_forward;           // projector
    float l_inv[9];                   // rows of the world -> local rotation
    float l_cot, l_scale;
    f3 l_irr;
    uint32_t capture_type, flags;
    int32_t filter_depth;
    uint32_t n_shapes;
    const NlosShape *shapes;
    const float *shape_pmf, *shape_cdf;
    const float *face_pmf, *face_cdf;  // per ORIGINAL triangle index, normalised within its shape
    const q4 *hg_tris;                 // 3 quads per ORIGINAL triangle: (p0, e1.x) (e1.yz, e2.xy) (e2.z, n)
    const q4 *hg_vn;                   // vertex normals of hidden meshes that have them (or null): mesh_sample_position
    // [W*H] scanned points (film order) | [1] the point the laser's axis hits (Single) | [laser_w*laser_h] the
    // illuminated points of an Exhaustive capture (transientnlospath.py:340-381)
    const q4 *targets;
    uint32_t film_w, film_h;
    uint32_t laser_w, laser_h;         // Exhaustive: illumination grid (film.laser_scan_width / _height), else 0
    float illum_tan;                   // Exhaustive without FORCE_EQUAL_GRIDS: tan(illumination_scan_fov / 2)
    float l_rot[9];                    // rows of the laser's local -> world rotation
    uint32_t camera_sensor;            // 1: the sensor is the scene's perspective camera (no nlos_capture_meter / relay wall)
    Camera cam;                        // ... that camera
    float inv_w, inv_h;                // 1 / film size (film sample of a pixel corner)
    uint32_t sensor_confocal;          // is_confocal capture meter: every sensor ray goes to sensor_target (nloscapturemeter.py:142)
    f3 sensor_target;
};

MTR_HD uint32_t nlos_target_count(const NlosConst &nc) { return nc.film_w * nc.film_h + 1u + nc.laser_w * nc.laser_h; }

MTR_HD f3 rect_point(f3 c, f3 du, f3 dv, float u, float v)
{
    const float a = fmaf(u, 2.0f, -1.0f), b = fmaf(v, 2.0f, -1.0f);
    return mk(fmaf(du.x, a, fmaf(dv.x, b, c.x)), fmaf(du.y, a, fmaf(dv.y, b, c.y)), fmaf(du.z, a, fmaf(dv.z, b, c.z)));
}

// [mitsuba3: Projector::sample_direction]: weight, ds.dist
MTR_HD f3 projector_sample(const NlosConst &nc, f3 p, float &dist)
{
    const f3 rel = p - nc.l_origin;
    const f3 loc = mk(dot(mk(nc.l_inv[0], nc.l_inv[1], nc.l_inv[2]), rel), dot(mk(nc.l_inv[3], nc.l_inv[4], nc.l_inv[5]), rel),
                      dot(mk(nc.l_inv[6], nc.l_inv[7], nc.l_inv[8]), rel));
    const float iz = 1.0f / loc.z;
    const float uvx = 0.5f - (0.5f * nc.l_cot) * (loc.x * iz), uvy = 0.5f - (0.5f * nc.l_cot) * (loc.y * iz);
    const bool ok = (uvx >= 0.0f) & (uvx <= 1.0f) & (uvy >= 0.0f) & (uvy <= 1.0f) & (loc.z > 0.0f);
    f3 d = nc.l_origin - p;
    dist = sqrtf(dot(d, d));
    d = d / dist;
    const float f = (kPi * nc.l_scale) * (iz * iz) / -dot(nc.l_forward, d);
    return ok ? mk(nc.l_irr.x * f, nc.l_irr.y * f, nc.l_irr.z * f) : mk(0, 0, 0);
}

// NLOSCaptureMeter.sample_ray for a film sample in [0,1)^2 (nloscapturemeter.py:136-180)
MTR_HD Ray nlos_sensor_ray(const NlosConst &nc, float sx, float sy)
{
    const float W = (float)nc.film_w, H = (float)nc.film_h;
    const float gx = (floorf(sx * W) + 0.5f) / W, gy = (floorf(sy * H) + 0.5f) / H;
    f3 target = rect_point(nc.w_center, nc.w_du, nc.w_dv, gx, gy);
    if (nc.sensor_confocal) target = nc.sensor_target;
    f3 dir = target - nc.sensor_origin;
    const float dist = sqrtf(dot(dir, dir));
    Ray r; r.o = nc.sensor_origin; r.d = dir / dist; r.tmax = kInf;
    return r;
}

// sensor.sample_ray at the film sample of pixel (x, y)'s corner (:296-308): the capture meter snaps it to the pixel centre
// (nloscapturemeter.py:146-149), a perspective camera shoots through it
MTR_HD Ray nlos_scan_ray(const NlosConst &nc, uint32_t x, uint32_t y)
{
    if (nc.camera_sensor) {
        RenderConst rc{};
        rc.inv_crop_w = nc.inv_w; rc.inv_crop_h = nc.inv_h; rc.off_x = 0.0f; rc.off_y = 0.0f;
        return camera_ray(nc.cam, rc, x, y, 0.0f, 0.0f);
    }
    return nlos_sensor_ray(nc, (float)x / (float)nc.film_w, (float)y / (float)nc.film_h);
}

// the rays of TransientNLOSPath.prepare (:295-381) in the order of NlosConst::targets
MTR_HD Ray nlos_prepare_ray(const NlosConst &nc, uint32_t i)
{
    const uint32_t n = nc.film_w * nc.film_h;
    Ray r;
    if (i < n) {                                                           // linspace(0,1,res,endpoint=False), meshgrid 'xy'
        const uint32_t y = i / nc.film_w, x = i - y * nc.film_w;
        return nlos_scan_ray(nc, x, y);
    }
    if (i == n) { r.o = nc.l_origin; r.d = nc.l_forward; r.tmax = kInf; return r; }
    const uint32_t j = i - n - 1u;
    if (nc.flags & MTR_NLOS_FORCE_EQUAL_GRIDS) {                           // laser_targets = sensor_targets (:344-346)
        const uint32_t y = j / nc.film_w, x = j - y * nc.film_w;
        return nlos_scan_ray(nc, x, y);
    }
    // dummy projector with illumination_scan_fov, one ray per grid point [mitsuba3: Projector::sample_ray with a
    // constant irradiance: uv = sample; near_p = sample_to_camera * (u, v, 0); d = to_world * normalize(near_p)]
    const uint32_t y = j / nc.laser_w, x = j - y * nc.laser_w;
    const float u = (float)x / (float)nc.laser_w, v = (float)y / (float)nc.laser_h;
    const f3 loc = normalize(mk((1.0f - 2.0f * u) * nc.illum_tan, (1.0f - 2.0f * v) * nc.illum_tan, 1.0f));
    r.o = nc.l_origin;
    r.d = mk(fmaf(nc.l_rot[0], loc.x, fmaf(nc.l_rot[1], loc.y, nc.l_rot[2] * loc.z)),
             fmaf(nc.l_rot[3], loc.x, fmaf(nc.l_rot[4], loc.y, nc.l_rot[5] * loc.z)),
             fmaf(nc.l_rot[6], loc.x, fmaf(nc.l_rot[7], loc.y, nc.l_rot[8] * loc.z)));
    r.tmax = kInf;
    return r;
}

MTR_HD void nlos_begin(Path &p, const NlosConst &nc, const Film &f, const RenderConst &rc, uint32_t pixel, uint32_t s)
{
    const uint32_t lane = pixel * rc.spp_total + s;
    const uint32_t py = pixel / f.crop_w, px = pixel - f.crop_w * py;
    p.px = px + f.crop_x; p.py = py + f.crop_y; p.lane = lane;
    p.rng = rng_seed(rc.seed, lane, rc.flags);
    const float j1 = rng_f32(p.rng), j2 = rng_f32(p.rng);
    const float sx = fmaf((float)p.px + j1, rc.inv_crop_w, rc.off_x), sy = fmaf((float)p.py + j2, rc.inv_crop_h, rc.off_y);
    p.ray = nc.camera_sensor ? camera_ray(nc.cam, rc, p.px, p.py, j1, j2) : nlos_sensor_ray(nc, sx, sy);
    p.beta = mk(1, 1, 1); p.L = mk(0, 0, 0); p.prev_p = mk(0, 0, 0);
    p.eta = 1.0f; p.dist = 0.0f; p.prev_pdf = 1.0f; p.depth = 0; p.prev_delta = 1;       // distance = ray.time = 0 (:718)
}

// si.spawn_ray_to(t) [mitsuba3: Interaction::spawn_ray_to]
MTR_HD Ray spawn_ray_to(f3 sp, f3 sn, f3 t)
{
    const f3 o = offset_point(sp, sn, t - sp);
    f3 dd = t - o;
    const float dist = sqrtf(dot(dd, dd));
    Ray r; r.o = o; r.d = dd / dist; r.tmax = dist * (1.0f - kShadowEps);
    return r;
}

// bsdf.eval (value * cos) of the smooth BSDFs: diffuse, and with EXT the GGX lobes; `albedo` = the reflectance at the hit
// (the material's constant, or its bitmap: material_albedo)
template <bool EXT, uint32_t TR = 0u>
MTR_HD f3 bsdf_eval_cos(const mtr_material &m, f3 albedo, f3 wi, f3 wo)
{
    const bool rough = lobes_on<EXT, TR>() && bsdf_is_rough(m.type);
    if (m.type != MTR_BSDF_DIFFUSE && !rough) return mk(0, 0, 0);
    if ((m.flags & MTR_MAT_TWOSIDED) && wi.z < 0.0f) { wi.z = -wi.z; wo.z = -wo.z; }
    if (rough) { f3 val; float pdf; rough_eval_pdf(m, albedo, wi, wo, val, pdf); return val; }
    if (!(wi.z > 0.0f && wo.z > 0.0f)) return mk(0, 0, 0);
    return mk((albedo.x * kInvPi) * wo.z, (albedo.y * kInvPi) * wo.z, (albedo.z * kInvPi) * wo.z);
}
template <bool EXT, uint32_t TR = 0u>
MTR_HD bool nlos_bsdf_smooth(const mtr_material &m) { return m.type == MTR_BSDF_DIFFUSE || (lobes_on<EXT, TR>() && bsdf_is_rough(m.type)); }

// emitter_nee_sample (transientnlospath.py:432-509); `depth` is the reference's argument (not the loop depth)
// `reload()` runs after every traversal: a caller that can re-read nc / film / rc (k_fused: from the kernarg segment, kernarg_copy)
// does so there instead of holding ~100 uniform values in (spilled) scalar registers across the walks — path_bounce's `refresh`
struct NoReload { MTR_HD void operator()() const {} };
// TR (here and below): scene traits (mtr_core.h) — kTrNoLobes takes the rough lobes' code out of the extended instantiations
template <bool EXT, uint32_t TR = 0u, class Stack, class Sink, class Reload = NoReload>
MTR_HD f3 nlos_emitter_nee(Path &p, const HitCtx &c, const mtr_material &mat, f3 albedo, f3 beta, float distance, uint32_t depth,
                           bool focus_laser, uint32_t laser, const SceneView &sc, const NlosConst &nc, const Film &film,
                           const RenderConst &rc, Stack &st, Sink &sink, BounceStats &stats, const Reload &reload = Reload())
{
    // visibility of the emitter origin (:441)
    const Ray sr = spawn_ray_to(c.sp, c.gn, nc.l_origin);
    stats.shadow++;
    const bool blocked = traverse<true>(sc, sr.o, sr.d, sr.tmax, st).prim >= 0;
    reload();
    if (blocked) return mk(0, 0, 0);
    (void)rng_f32(p.rng); (void)rng_f32(p.rng);                      // sampler.next_2d(active_e): only visible lanes draw
    float ds_dist;
    f3 w;
    if (focus_laser && nc.capture_type != MTR_CAPTURE_SINGLE) {      // Confocal or Exhaustive, :448-458
        const f3 rel = nc.l_origin - c.sp;
        const float dist_e = sqrtf(dot(rel, rel));
        w = projector_sample(nc, fma3(nc.l_forward, dist_e, nc.l_origin), ds_dist);
    } else {
        w = projector_sample(nc, c.sp, ds_dist);
    }
    const f3 dirn = normalize(nc.l_origin - c.sp);                   // :483
    const f3 wo = mk(dot(dirn, c.ss), dot(dirn, c.stt), dot(dirn, c.sn));
    const f3 bv = bsdf_eval_cos<EXT, TR>(mat, albedo, c.wi, wo);
    if (nc.filter_depth != -1 && depth != (uint32_t)nc.filter_depth) return mk(0, 0, 0);      // :489-490
    if ((nc.flags & MTR_NLOS_DISCARD_DIRECT) && !(depth > 2)) return mk(0, 0, 0);             // :491-492
    const f3 Lr = mk((beta.x * bv.x) * w.x, (beta.y * bv.y) * w.y, (beta.z * bv.z) * w.z);    // :493
    if (nc.flags & MTR_NLOS_ACCOUNT_FIRST_LAST) distance += ds_dist * p.eta;                   // :497-498
    const uint32_t fx = p.px - film.crop_x, fy = p.py - film.crop_y;
    const float vr = Lr.x * rc.sample_scale, vg = Lr.y * rc.sample_scale, vb = Lr.z * rc.sample_scale;
    if ((fx < film.width) & (fy < film.height) && (vr != 0.0f || vg != 0.0f || vb != 0.0f)) {
        const int32_t bin = film_row_bin(film, distance, laser);                              // [laser_x][laser_y][t], :499-507
        if (bin >= 0) sink.splat(fx, fy, (uint32_t)bin, vr, vg, vb, distance, p.depth, 1u);
    }
    return Lr;
}

// emitter_laser_targets_sample (:511-564)
template <bool EXT, uint32_t TR = 0u, class Stack, class Sink, class Reload = NoReload>
MTR_HD f3 nlos_laser_targets(Path &p, const HitCtx &c, const mtr_material &mat, f3 albedo, f3 lt, uint32_t depth, uint32_t laser,
                             const SceneView &sc, const NlosConst &nc, const Film &film, const RenderConst &rc,
                             Stack &st, Sink &sink, BounceStats &stats, const Reload &reload = Reload())
{
    f3 dd = lt - c.sp;
    const float dl = sqrtf(dot(dd, dd));
    dd = dd / dl;
    Ray rb = spawn_ray_to(c.sp, c.gn, lt);
    stats.shadow++;
    if (traverse<true>(sc, rb.o, rb.d, rb.tmax, st).prim >= 0) { reload(); return mk(0, 0, 0); }             // :528
    const f3 wo = mk(dot(dd, c.ss), dot(dd, c.stt), dot(dd, c.sn));
    const f3 bs = bsdf_eval_cos<EXT, TR>(mat, albedo, c.wi, wo);                              // :531-533
    const Hit h2 = traverse<false>(sc, rb.o, rb.d, kInf, st);                                  // :535-537
    reload();
    stats.closest++;
    if (h2.prim < 0) return mk(0, 0, 0);
    if (!(bs.x > kDrEps || bs.y > kDrEps || bs.z > kDrEps)) return mk(0, 0, 0);               // :539-540
    const HitCtx c2 = hit_ctx<EXT>(sc, rb.d, h2);
    const f3 md = -dd;
    const float wlz = dot(md, c2.sn);                                                          // cos_theta(si_bsdf.to_local(-d))
    if (!(wlz > 0.0f)) return mk(0, 0, 0);                                                     // :543
    const float pdf_ls = (dl * dl) / wlz;                                                      // :546-551
    const f3 b2 = mk(p.beta.x * (bs.x / pdf_ls), p.beta.y * (bs.y / pdf_ls), p.beta.z * (bs.z / pdf_ls));
    return nlos_emitter_nee<EXT, TR>(p, c2, sc.mats[c2.mat], material_albedo<EXT>(sc, sc.mats[c2.mat], h2), b2, p.dist + dl * p.eta, depth + 1, true,
                                 laser, sc, nc, film, rc, st, sink, stats, reload);
}

// hidden_geometry_sample (:637-670) incl. _sample_hidden_geometry_position (:385-430)
template <bool EXT, uint32_t TR = 0u>
MTR_HD BsdfSample nlos_hidden_geometry(const HitCtx &c, const mtr_material &mat, f3 albedo, float ua, float ub, const NlosConst &nc)
{
    BsdfSample bs;
    bs.wo = mk(0, 0, 0); bs.pdf = 0.0f; bs.eta = 1.0f; bs.delta = false; bs.w = mk(0, 0, 0);
    float reused, spmf;
    const uint32_t s = distr_sample_reuse(nc.shape_cdf, nc.shape_pmf, nc.n_shapes, ua, reused, spmf);
    const NlosShape &S = nc.shapes[s];
    f3 pp, pn;
    if (S.is_rect) {
        pp = rect_point(ld3(S.center), ld3(S.du), ld3(S.dv), reused, ub);
        pn = ld3(S.n);
    } else {                                               // [mitsuba3: Mesh::sample_position]
        mesh_sample_position(nc.hg_tris, nc.face_cdf, nc.face_pmf, S.first_tri, S.n_tris, reused, ub, pp, pn, EXT ? nc.hg_vn : nullptr);
    }
    const float ppdf = S.inv_area * spmf;
    f3 dd = pp - c.sp;
    const float dist = sqrtf(dot(dd, dd));
    dd = dd / dist;
    const float cos_i = dot(c.gn, dd), cos_g = dot(pn, -dd);               // si.n: the geometric normal
    const f3 wo = mk(dot(dd, c.ss), dot(dd, c.stt), dot(dd, c.sn));
    bs.wo = wo;
    bs.pdf = ppdf * (dist * dist) / fabsf(cos_g);
    if (!(cos_i > kDrEps && cos_g > kDrEps)) return bs;
    if (!(bs.pdf > kDrEps)) return bs;
    const f3 val = bsdf_eval_cos<EXT, TR>(mat, albedo, c.wi, wo);
    bs.w = mk(val.x / bs.pdf, val.y / bs.pdf, val.z / bs.pdf);
    return bs;
}

// the laser target of this pixel: its own scanned point (Confocal, :337-339, :585-589) or the single point
MTR_HD f3 nlos_laser_target(const NlosConst &nc, uint32_t px, uint32_t py)
{
    const size_t i = (nc.capture_type == MTR_CAPTURE_CONFOCAL) ? (size_t)py * nc.film_w + px : (size_t)nc.film_w * nc.film_h;
    const q4 t = nc.targets[i];
    return mk(t.x, t.y, t.z);
}

// One iteration of TransientNLOSPath.sample (:740-918).  Returns active_next.
template <bool EXT = false, uint32_t TR = 0u, class Stack, class Sink, class Reload = NoReload>
MTR_HD bool nlos_bounce(Path &p, const SceneView &sc, const NlosConst &nc, const Film &film, const RenderConst &rc,
                        Stack &st, Sink &sink, BounceStats &stats, const Reload &reload = Reload())
{
    const Hit h = traverse<false>(sc, p.ray.o, p.ray.d, p.ray.tmax, st);
    reload();
    stats.closest++;
    const bool valid = h.prim >= 0;
    if ((nc.flags & MTR_NLOS_ACCOUNT_FIRST_LAST) || p.depth > 0) p.dist += h.t * p.eta;          // :751-752
    bool active_next = ((p.depth + 1u) < rc.max_depth) & valid;                                   // :782
    f3 Lr = mk(0, 0, 0);
    HitCtx c;
    c.sp = mk(0, 0, 0); c.sn = mk(0, 0, 1); c.gn = mk(0, 0, 1); c.ss = mk(1, 0, 0); c.stt = mk(0, 1, 0); c.wi = mk(0, 0, 0); c.mat = 0; c.em_plus1 = 0;
    if (valid) c = hit_ctx<EXT>(sc, p.ray.d, h);
    const mtr_material &mat = sc.mats[c.mat];
    const f3 albedo = valid ? material_albedo<EXT>(sc, mat, h) : mk(0, 0, 0);
    // the only emitter is the projector (not a surface): Le = 0 (:757-777)
    if (active_next && nlos_bsdf_smooth<EXT, TR>(mat)) {                                 // active_em :785-786
        if ((nc.flags & MTR_NLOS_LASER_SAMPLING) && nc.capture_type == MTR_CAPTURE_EXHAUSTIVE) {
            // every illuminated point in turn (:597-621); target i lands in film cell laser_x = i / Ly, laser_y = i % Ly,
            // i.e. row offset i * T; the steady estimate is the mean over the grid
            const uint32_t nl = nc.laser_w * nc.laser_h, t0 = nc.film_w * nc.film_h + 1u;
            for (uint32_t i = 0; i < nl; ++i) {
                const q4 t = nc.targets[t0 + i];
                const f3 li = nlos_laser_targets<EXT, TR>(p, c, mat, albedo, mk(t.x, t.y, t.z), p.depth + 1u, i, sc, nc, film, rc, st, sink, stats, reload);
                Lr = mk(Lr.x + li.x, Lr.y + li.y, Lr.z + li.z);
            }
            const float nlf = (float)nl;
            Lr = mk(Lr.x / nlf, Lr.y / nlf, Lr.z / nlf);
        } else if (nc.flags & MTR_NLOS_LASER_SAMPLING)                                            // emitter_laser_sample: depth + 1
            Lr = nlos_laser_targets<EXT, TR>(p, c, mat, albedo, nlos_laser_target(nc, p.px, p.py), p.depth + 1u, 0u, sc, nc, film, rc, st, sink, stats, reload);
        else
            Lr = nlos_emitter_nee<EXT, TR>(p, c, mat, albedo, p.beta, p.dist, p.depth, false, 0u, sc, nc, film, rc, st, sink, stats, reload);
    }
    // hidden-geometry / BSDF sampling (:797-833)
    const bool hg = (nc.flags & MTR_NLOS_HG_SAMPLING) != 0;
    bool do_hg = hg;
    float pdf_method = 1.0f;
    if (hg && (nc.flags & MTR_NLOS_HG_RROULETTE)) { do_hg = rng_f32(p.rng) < 0.5f; pdf_method = 0.5f; }   // :801
    (void)rng_f32(p.rng);                                                                         // :814 next_1d (unused)
    const float a2a = rng_f32(p.rng), a2b = rng_f32(p.rng);
    const float b1 = rng_f32(p.rng), b2a = rng_f32(p.rng), b2b = rng_f32(p.rng);                  // :820
    const float rr_u = rng_f32(p.rng);                                                            // :857
    BsdfSample bs;
    bs.wo = mk(0, 0, 0); bs.pdf = 0.0f; bs.eta = 1.0f; bs.delta = false; bs.w = mk(0, 0, 0);
    if (active_next) {
        if (do_hg) bs = nlos_hidden_geometry<EXT, TR>(c, mat, albedo, a2a, a2b, nc);
        else bs = bsdf_sample<EXT, TR & kTrNoLobes>(mat, c.wi, b1, b2a, b2b, albedo);
        const f3 wo_w = mk(fmaf(c.sn.x, bs.wo.z, fmaf(c.stt.x, bs.wo.y, c.ss.x * bs.wo.x)),
                           fmaf(c.sn.y, bs.wo.z, fmaf(c.stt.y, bs.wo.y, c.ss.y * bs.wo.x)),
                           fmaf(c.sn.z, bs.wo.z, fmaf(c.stt.z, bs.wo.y, c.ss.z * bs.wo.x)));
        p.ray.o = offset_point(c.sp, c.gn, wo_w); p.ray.d = wo_w; p.ray.tmax = kInf;
    }
    p.L = mk(p.L.x + Lr.x, p.L.y + Lr.y, p.L.z + Lr.z);
    p.eta *= bs.eta;
    p.beta = mk((p.beta.x * bs.w.x) / pdf_method, (p.beta.y * bs.w.y) / pdf_method, (p.beta.z * bs.w.z) / pdf_method);   // :833
    const float bmax = max3(p.beta.x, p.beta.y, p.beta.z);
    active_next &= (bmax != 0.0f);
    const float rr_prob = fminf(bmax * (p.eta * p.eta), 0.95f);
    active_next &= rr_prob > 0.0f;
    const bool rr_active = p.depth >= rc.rr_depth;
    if (rr_active) { const float inv = rr_prob > 0.0f ? 1.0f / rr_prob : 0.0f; p.beta = p.beta * inv; }
    active_next &= (!rr_active) | (rr_u < rr_prob);
    if (valid) p.depth += 1;
    return active_next;
}

} // namespace mtr
