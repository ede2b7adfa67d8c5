/*
 * mtr_oracle.c — CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A plain-C, one-lane-at-a-time restatement of the reference hot path
 *   TransientADIntegrator.render      mitransient/integrators/common.py:122-213
 *   TransientADIntegrator.prepare     mitransient/integrators/common.py:32-85
 *   TransientPath.sample              mitransient/integrators/transientpath.py:88-326
 *   TransientHDRFilm.add_transient_data / develop
 *                                     mitransient/films/transient_hdr_film.py:210-276
 *   TransientImageBlock.put_/accum    mitransient/render/transient_image_block.py:79-151
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this file's shared object.  The product (mitransient_amd/) never does.
 *
 * PARITY UNPINNED.  The arithmetic below the reference's Python (ray/scene
 * intersection, BSDFs, emitter sampling, the PCG32 sampler, the perspective
 * sensor) lives in the un-vendored third-party dependency
 *   mitsuba >=3.6.0,<3.9.0 (setup.py:21, version.py:5-7; "latest" 3.8.0) + its pinned drjit,
 * which is absent from /root/reference and cannot be installed here (no
 * network, no wheel).  Those parts restate Mitsuba 3's published algorithms
 * (cited inline as [mitsuba3: file]); the reference holds no golden vectors
 * for this path (tests/integration/test_nlos.py:117-118 asserts shapes only).
 * What IS pinned: the official PCG32 known-answer stream, the f32 bin mapping
 * of transient_hdr_film.py:263-265, the energy identity transient.sum(2) ==
 * steady (examples/transient-nlos/1-simple-nlos-scenes.ipynb, md cell 8), an
 * analytic direct-illumination quadrature (tests/) — and, at FIGURE precision,
 * outputs of the reference itself (tests/test_reference_figures.py, fixtures
 * decoded by tests/golden/make_reference_figures.py): the figures embedded in
 * its notebooks (NLOS frames and a pixel's time response with absolute sums
 * within 0.1 - 3.5 %, the Cornell box's iso-time bands, the phasor film) and
 * the README's own renders — the steady staircase (mean linear colour within
 * 9 % here, 1 - 3 % for the product at full sample counts), the steady Cornell
 * box, and the staircase's transient video, whose frame k is time bin k + 20 of
 * this oracle's render.  No sample-for-sample comparison with Mitsuba exists:
 * in that strict sense parity stays unpinned.
 *
 * Numerics contract shared with the HIP path: IEEE f32,
 * no contraction (-ffp-contract=off), fmaf() only where written, 1/x and
 * sqrtf correctly rounded, sin/cos by the fixed polynomials below.
 */
#include "mtr_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { float x, y, z; } v3;

static inline v3 V(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vscale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }
/* [drjit: dot() of a 3-vector is an fma chain] */
static inline float vdot(v3 a, v3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
/* [drjit: cross() = fmsub(a.yzx, b.zxy, a.zxy * b.yzx)] */
static inline v3 vcross(v3 a, v3 b)
{
    return V(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}
/* vector / scalar: ONE correctly rounded reciprocal, three multiplies [drjit: a / s == a * rcp(s)] */
static inline v3 vdivs(v3 a, float s) { float r = 1.0f / s; return vscale(a, r); }
static inline v3 vnormalize(v3 a) { return vdivs(a, sqrtf(vdot(a, a))); }
/* a*s + b, per component */
static inline v3 vfma(v3 a, float s, v3 b) { return V(fmaf(a.x, s, b.x), fmaf(a.y, s, b.y), fmaf(a.z, s, b.z)); }
static inline float mulsign(float x, float s) { return (s < 0.0f || (s == 0.0f && signbit(s))) ? -x : x; }

/* ------------------------------------------------------------------ */
/* Independent sampler  [mitsuba3: src/samplers/independent.cpp,       */
/* include/mitsuba/core/random.h sample_tea_32, drjit/random.h PCG32]  */
/* reference call sites: common.py:52, transientpath.py:193,223-224,256 */
/* ------------------------------------------------------------------ */
typedef struct { uint64_t state, inc; } pcg32;

static void tea32(uint32_t *v0, uint32_t *v1, int rounds)
{
    uint32_t a = *v0, b = *v1, sum = 0;
    for (int i = 0; i < rounds; ++i) {
        sum += 0x9e3779b9u;
        a += ((b << 4) + 0xa341316cu) ^ (b + sum) ^ ((b >> 5) + 0xc8013ea4u);
        b += ((a << 4) + 0xad90777du) ^ (a + sum) ^ ((a >> 5) + 0x7e95761eu);
    }
    *v0 = a; *v1 = b;
}
static uint32_t pcg32_next_u32(pcg32 *r)
{
    uint64_t old = r->state;
    r->state = old * 0x5851f42d4c957f2dULL + r->inc;
    uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u);
    uint32_t rot = (uint32_t)(old >> 59u);
    return (xs >> rot) | (xs << ((~rot + 1u) & 31u));
}
static void pcg32_seed(pcg32 *r, uint64_t initstate, uint64_t initseq)
{
    r->state = 0u;
    r->inc = (initseq << 1u) | 1u;
    pcg32_next_u32(r);
    r->state += initstate;
    pcg32_next_u32(r);
}
static float pcg32_next_f32(pcg32 *r)
{
    union { uint32_t u; float f; } x;
    x.u = (pcg32_next_u32(r) >> 9) | 0x3f800000u;
    return x.f - 1.0f;
}
/* sampler.seed(seed, wavefront_size): per-lane stream from TEA(seed, lane) */
static void sampler_seed_v(pcg32 *r, uint32_t seed_value, uint32_t lane, uint32_t flags)
{
    const int seq_plus_lane = (flags & MTR_FLAG_PCG_INITSEQ_PLUS_LANE) != 0;
    uint32_t v0 = seed_value, v1 = lane;
    tea32(&v0, &v1, 4);
    if (flags & MTR_FLAG_PCG_TEA64) {
        /* third reading [upstream-unverified]: m_rng.seed(1, sample_tea_64(seed, idx), sample_tea_64(idx, seed)) with
         * sample_tea_64(a, b) = v0 + (v1 << 32) of sample_tea_32(a, b): 64-bit state and stream words */
        uint32_t w0 = lane, w1 = seed_value;
        tea32(&w0, &w1, 4);
        pcg32_seed(r, (uint64_t)v0 + ((uint64_t)v1 << 32), (uint64_t)w0 + ((uint64_t)w1 << 32));
        return;
    }
    /* [drjit: PCG32::seed(size, initstate, initseq)] inc = ((initseq + arange(size)) << 1) | 1; mitsuba's sampler seeds with
     * size = 1 after the scramble (seq_plus_lane = 0); the other reading is kept switchable (MTR_FLAG_PCG_INITSEQ_PLUS_LANE) */
    pcg32_seed(r, (uint64_t)v0, (uint64_t)v1 + (seq_plus_lane ? (uint64_t)lane : 0u));
}
static void sampler_seed(pcg32 *r, uint32_t seed_value, uint32_t lane) { sampler_seed_v(r, seed_value, lane, 0); }

/* exported KAT hooks */
void orc_pcg32_stream(uint64_t initstate, uint64_t initseq, uint32_t n, uint32_t *out_u32, float *out_f32)
{
    pcg32 r; pcg32_seed(&r, initstate, initseq);
    pcg32 r2 = r;
    for (uint32_t i = 0; i < n; ++i) {
        if (out_u32) out_u32[i] = pcg32_next_u32(&r);
        if (out_f32) out_f32[i] = pcg32_next_f32(&r2);
    }
}
void orc_sampler_stream(uint32_t seed_value, uint32_t lane, uint32_t n, float *out)
{
    pcg32 r; sampler_seed(&r, seed_value, lane);
    for (uint32_t i = 0; i < n; ++i) out[i] = pcg32_next_f32(&r);
}
void orc_tea32(uint32_t v0, uint32_t v1, int rounds, uint32_t *out2)
{
    tea32(&v0, &v1, rounds); out2[0] = v0; out2[1] = v1;
}

/* ------------------------------------------------------------------ */
/* sin/cos on [-pi/4, pi/4]: fixed minimax polynomials (Cephes sinf /  */
/* cosf kernels); part of the numerics contract.                       */
/* ------------------------------------------------------------------ */
static void sincos_q(float x, float *s, float *c)
{
    float z = x * x;
    float ps = fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
    *s = fmaf(x * z, ps, x);
    float pc = fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
    *c = fmaf(z * z, pc, fmaf(-0.5f, z, 1.0f));
}
void orc_sincos_q(float x, float *s, float *c) { sincos_q(x, s, c); }

/* One frequency of a phasor_hdr_film contribution (mitransient/render/phasor_image_block.py:49-56):
 *   phase = fmod(-2 pi f opl, 2 pi), fmod(x, y) = x - y * floor(x / y); values spec * cos(phase), spec * sin(phase).
 * All f32.  Dr.Jit's sincos cannot be reproduced bit for bit; the contract here: reduce to a multiple of pi/2 with the
 * two-constant Cody-Waite split (fma), then the quarter-range polynomials above. */
static void phasor_term(float freq, float opl, float *c, float *s)
{
    const float x = (-6.283185307179586f * freq) * opl, y = 6.283185307179586f;
    const float phase = x - y * floorf(x / y);
    const float k = floorf(fmaf(phase, 0.6366197723675814f, 0.5f));
    float r = fmaf(-k, 1.5707962512969971f, phase);
    r = fmaf(-k, 7.549789415861596e-08f, r);
    float sq, cq;
    sincos_q(r, &sq, &cq);
    const uint32_t q = (uint32_t)(int32_t)k & 3u;
    *s = (q == 0u) ? sq : (q == 1u) ? cq : (q == 2u) ? -sq : -cq;
    *c = (q == 0u) ? cq : (q == 1u) ? -sq : (q == 2u) ? -cq : sq;
}
void orc_phasor_term(float freq, float opl, float *c, float *s) { phasor_term(freq, opl, c, s); }

#define ORC_PI       3.14159265358979323846f
#define ORC_INV_PI   0.31830988618379067154f
#define ORC_RAY_EPS  (1500.0f * 5.9604644775390625e-8f)          /* [mitsuba3: math::RayEpsilon = Epsilon*1500, Epsilon<float> = 2^-24] */
#define ORC_SHADOW_EPS (ORC_RAY_EPS * 10.0f)                      /* [mitsuba3: math::ShadowEpsilon] */

/* [mitsuba3: warp::square_to_uniform_disk_concentric] */
static void square_to_disk(float u1, float u2, float *px, float *py)
{
    float x = fmaf(2.0f, u1, -1.0f), y = fmaf(2.0f, u2, -1.0f);
    int is_zero = (x == 0.0f && y == 0.0f);
    int q13 = fabsf(x) < fabsf(y);
    float r = q13 ? y : x, rp = q13 ? x : y;
    float phi = (0.25f * ORC_PI) * rp / r;       /* in [-pi/4, pi/4] */
    if (is_zero) phi = 0.0f;
    float s, c;
    sincos_q(phi, &s, &c);
    if (q13) { float t = s; s = c; c = t; }      /* sincos(pi/2 - phi) */
    *px = r * c; *py = r * s;
}
/* [mitsuba3: warp::square_to_cosine_hemisphere] */
static v3 square_to_cos_hemi(float u1, float u2)
{
    float px, py;
    square_to_disk(u1, u2, &px, &py);
    float zz = 1.0f - fmaf(px, px, py * py);
    float z = sqrtf(zz > 0.0f ? zz : 0.0f);
    return V(px, py, z);
}
void orc_square_to_cos_hemi(float u1, float u2, float *out3)
{
    v3 w = square_to_cos_hemi(u1, u2); out3[0] = w.x; out3[1] = w.y; out3[2] = w.z;
}

/* ------------------------------------------------------------------ */
/* Scene, derived per-triangle data                                    */
/* ------------------------------------------------------------------ */
typedef struct {
    v3 p0, e1, e2;     /* [mitsuba3: Mesh::ray_intersect_triangle] */
    v3 n, s, t;        /* geometric normal; frame s,t of a flat-shaded triangle [mitsuba3: SurfaceInteraction::initialize_sh_frame] */
    v3 dp_du;          /* the tangent direction the frame is built from (smooth-shaded triangles rebuild it at the hit) */
    int smooth;        /* mtr_scene_desc.tri_normals holds vertex normals for this triangle */
    /* analytic `rectangle` [mitsuba3: src/shapes/rectangle.cpp]: kind 1 = the primitive (first of its two carrier
     * triangles), kind 2 = the second carrier (never intersected), kind 0 = a mesh triangle */
    int kind;
    v3 qc, qdu, qdv;   /* to_world * (0,0,0), to_world * (1,0,0) - c, to_world * (0,1,0) - c */
    float rx[4], ry[4], rz[4];   /* rows of to_object (3 x 4 affine): local = R * p + T */
} orc_tri;

typedef struct { v3 lo, hi; int left, right, first, count; } orc_node;

typedef struct {
    const mtr_scene_desc *d;
    orc_tri *tris;
    /* emitters: derived n, inv_area */
    v3 *em_n; float *em_inv_area;
    float *em_face_pmf, *em_face_cdf;          /* mesh emitters: per triangle (original index), normalised within the emitter */
    /* own small BVH (median split), only used when use_bvh != 0 */
    orc_node *nodes; int n_nodes; int *tri_order;
} orc_scene;

static double orc_tri_area_d(const float *v)
{
    double e1[3] = { (double)v[3] - v[0], (double)v[4] - v[1], (double)v[5] - v[2] };
    double e2[3] = { (double)v[6] - v[0], (double)v[7] - v[1], (double)v[8] - v[2] };
    double c[3] = { e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0] };
    return 0.5 * sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
}
static uint32_t distr_sample_reuse(const float *cdf, const float *pmf, uint32_t n, float value, float *reused, float *pmf_out);

/* [mitsuba3: coordinate_system(n)] (Duff et al. 2017), first vector only:
 *   sign = copysign(1, n.z); a = -rcp(sign + n.z); b = n.x * n.y * a;
 *   s = (mulsign(sqr(n.x) * a, n.z) + 1, mulsign(b, n.z), mulsign_neg(n.x, n.z)) */
static v3 coordinate_system_s(v3 n)
{
    float sign = copysignf(1.0f, n.z);
    float a = -(1.0f / (sign + n.z));
    float b = (n.x * n.y) * a;
    return V(mulsign((n.x * n.x) * a, n.z) + 1.0f, mulsign(b, n.z), mulsign(-n.x, n.z));
}
/* [mitsuba3: SurfaceInteraction::initialize_sh_frame] s = normalize(fnmadd(n, dot(n, dp_du), dp_du)); t = cross(n, s) */
static void sh_frame_from(v3 n, v3 dp_du, v3 *s, v3 *t)
{
    float dn = vdot(n, dp_du);
    *s = vnormalize(V(fmaf(-n.x, dn, dp_du.x), fmaf(-n.y, dn, dp_du.y), fmaf(-n.z, dn, dp_du.z)));
    *t = vcross(n, *s);
}
/* to_object of a rectangle from (c, du, dv): the rows of the inverse of [du dv n^ | c], n^ = normalize(du x dv)
 * (mitsuba inverts the full 4x4 to_world; the third column only scales o'.z and d'.z alike, so t = -o'.z / d'.z and the
 * local x, y are the same up to rounding).  f64, rounded to f32 — shared numerics contract with the HIP library. */
static void rect_to_object(const float c[3], const float du[3], const float dv[3], float rx[4], float ry[4], float rz[4])
{
    double a[3] = { du[0], du[1], du[2] }, b[3] = { dv[0], dv[1], dv[2] }, o[3] = { c[0], c[1], c[2] };
    double n[3] = { a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0] };
    double ln = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    n[0] /= ln; n[1] /= ln; n[2] /= ln;
    double bn[3] = { b[1] * n[2] - b[2] * n[1], b[2] * n[0] - b[0] * n[2], b[0] * n[1] - b[1] * n[0] };   /* dv x n */
    double na[3] = { n[1] * a[2] - n[2] * a[1], n[2] * a[0] - n[0] * a[2], n[0] * a[1] - n[1] * a[0] };   /* n x du */
    double da = a[0] * bn[0] + a[1] * bn[1] + a[2] * bn[2], db = b[0] * na[0] + b[1] * na[1] + b[2] * na[2];
    double X[3] = { bn[0] / da, bn[1] / da, bn[2] / da }, Y[3] = { na[0] / db, na[1] / db, na[2] / db };
    for (int k = 0; k < 3; ++k) { rx[k] = (float)X[k]; ry[k] = (float)Y[k]; rz[k] = (float)n[k]; }
    rx[3] = (float)(-(X[0] * o[0] + X[1] * o[1] + X[2] * o[2]));
    ry[3] = (float)(-(Y[0] * o[0] + Y[1] * o[1] + Y[2] * o[2]));
    rz[3] = (float)(-(n[0] * o[0] + n[1] * o[1] + n[2] * o[2]));
}

static void build_tris(orc_scene *sc)
{
    const mtr_scene_desc *d = sc->d;
    sc->tris = (orc_tri *)calloc(d->n_tris ? d->n_tris : 1, sizeof(orc_tri));
    for (uint32_t i = 0; i < d->n_tris; ++i) {
        const float *v = d->tri_verts + 9 * (size_t)i;
        orc_tri *T = &sc->tris[i];
        v3 p0 = V(v[0], v[1], v[2]), p1 = V(v[3], v[4], v[5]), p2 = V(v[6], v[7], v[8]);
        T->p0 = p0; T->e1 = vsub(p1, p0); T->e2 = vsub(p2, p0);
        T->n = vnormalize(vcross(T->e1, T->e2));
        /* [mitsuba3: Mesh::compute_surface_interaction] dp_du from the UV parameterisation when the mesh has texture
         * coordinates and they are not degenerate, coordinate_system(n) otherwise */
        v3 dp_du = coordinate_system_s(T->n);
        if (d->tri_uv) {
            const float *uv = d->tri_uv + 6 * (size_t)i;
            float duv0x = uv[2] - uv[0], duv0y = uv[3] - uv[1], duv1x = uv[4] - uv[0], duv1y = uv[5] - uv[1];
            float det = fmaf(duv0x, duv1y, -(duv0y * duv1x));               /* fmsub(duv0.x, duv1.y, duv0.y * duv1.x) */
            if (det != 0.0f) {
                float inv_det = 1.0f / det;
                dp_du = V(fmaf(duv1y, T->e1.x, -(duv0y * T->e2.x)) * inv_det,   /* fmsub(duv1.y, dp0, duv0.y * dp1) * inv_det */
                          fmaf(duv1y, T->e1.y, -(duv0y * T->e2.y)) * inv_det,
                          fmaf(duv1y, T->e1.z, -(duv0y * T->e2.z)) * inv_det);
            }
        }
        sh_frame_from(T->n, dp_du, &T->s, &T->t);
        T->dp_du = dp_du; T->smooth = 0;
        if (d->tri_normals) for (int k = 0; k < 9; ++k) if (d->tri_normals[9 * (size_t)i + k] != 0.0f) T->smooth = 1;
    }
    /* analytic rectangles [mitsuba3: Rectangle::update / compute_surface_interaction]: one primitive, one frame */
    for (uint32_t k = 0; k < d->n_shapes && d->shapes; ++k) {
        const mtr_shape *S = &d->shapes[k];
        if (!S->is_rectangle || S->n_tris != 2 || (uint64_t)S->first_tri + 2 > d->n_tris) continue;
        orc_tri *Q = &sc->tris[S->first_tri];
        Q->kind = 1; sc->tris[S->first_tri + 1].kind = 2;
        Q->qc = V(S->center[0], S->center[1], S->center[2]);
        Q->qdu = V(S->du[0], S->du[1], S->du[2]); Q->qdv = V(S->dv[0], S->dv[1], S->dv[2]);
        Q->n = vnormalize(vcross(Q->qdu, Q->qdv));                          /* normalize(to_world * Normal3f(0, 0, 1)) */
        if (S->is_rectangle & MTR_RECT_FLIP_NORMALS) Q->n = V(-Q->n.x, -Q->n.y, -Q->n.z);   /* [Rectangle: flip_normals] */
        sh_frame_from(Q->n, Q->qdu, &Q->s, &Q->t);                          /* dp_du = to_world * (2, 0, 0): same direction */
        rect_to_object(S->center, S->du, S->dv, Q->rx, Q->ry, Q->rz);
        sc->tris[S->first_tri + 1].n = Q->n; sc->tris[S->first_tri + 1].s = Q->s; sc->tris[S->first_tri + 1].t = Q->t;
    }
    sc->em_n = (v3 *)calloc(d->n_emitters ? d->n_emitters : 1, sizeof(v3));
    sc->em_inv_area = (float *)calloc(d->n_emitters ? d->n_emitters : 1, sizeof(float));
    sc->em_face_pmf = (float *)calloc(d->n_tris ? d->n_tris : 1, sizeof(float));
    sc->em_face_cdf = (float *)calloc(d->n_tris ? d->n_tris : 1, sizeof(float));
    for (uint32_t i = 0; i < d->n_emitters; ++i) {
        const mtr_emitter *e = &d->emitters[i];
        if (e->is_mesh) {                       /* [mitsuba3: Mesh::surface_area / m_area_pmf] f64 sums, stored f32 */
            double a = 0.0, acc = 0.0;
            for (uint32_t t = 0; t < e->n_tris; ++t) a += orc_tri_area_d(d->tri_verts + 9 * (size_t)(e->first_tri + t));
            for (uint32_t t = 0; t < e->n_tris; ++t) {
                double at = orc_tri_area_d(d->tri_verts + 9 * (size_t)(e->first_tri + t));
                acc += at;
                sc->em_face_pmf[e->first_tri + t] = (float)(at / a);
                sc->em_face_cdf[e->first_tri + t] = (float)(acc / a);
            }
            sc->em_inv_area[i] = (float)(1.0 / a);
            continue;
        }
        v3 du = V(e->du[0], e->du[1], e->du[2]), dv = V(e->dv[0], e->dv[1], e->dv[2]);
        v3 c = vcross(du, dv);
        float len = sqrtf(vdot(c, c));
        sc->em_n[i] = vdivs(c, len);
        if (e->flip_normals) sc->em_n[i] = V(-sc->em_n[i].x, -sc->em_n[i].y, -sc->em_n[i].z);
        /* [mitsuba3: Rectangle: surface_area = |cross(dp_du, dp_dv)|, dp_du = to_world*(2,0,0)] */
        sc->em_inv_area[i] = 1.0f / (4.0f * len);
    }
}

/* ---- the oracle's own BVH: plain median split on the largest axis ---- */
static void tri_bounds(const float *v, v3 *lo, v3 *hi)
{
    *lo = V(INFINITY, INFINITY, INFINITY); *hi = V(-INFINITY, -INFINITY, -INFINITY);
    for (int k = 0; k < 3; ++k) {
        float x = v[3 * k], y = v[3 * k + 1], z = v[3 * k + 2];
        if (x < lo->x) lo->x = x; if (y < lo->y) lo->y = y; if (z < lo->z) lo->z = z;
        if (x > hi->x) hi->x = x; if (y > hi->y) hi->y = y; if (z > hi->z) hi->z = z;
    }
}
static const float *g_sort_verts; static int g_sort_axis;
static int cmp_centroid(const void *a, const void *b)
{
    int ia = *(const int *)a, ib = *(const int *)b;
    const float *va = g_sort_verts + 9 * (size_t)ia, *vb = g_sort_verts + 9 * (size_t)ib;
    float ca = va[g_sort_axis] + va[3 + g_sort_axis] + va[6 + g_sort_axis];
    float cb = vb[g_sort_axis] + vb[3 + g_sort_axis] + vb[6 + g_sort_axis];
    return (ca < cb) ? -1 : (ca > cb) ? 1 : (ia - ib);
}
static int build_node(orc_scene *sc, int first, int count)
{
    int idx = sc->n_nodes++;
    orc_node *N = &sc->nodes[idx];
    v3 lo = V(INFINITY, INFINITY, INFINITY), hi = V(-INFINITY, -INFINITY, -INFINITY);
    for (int i = first; i < first + count; ++i) {
        int p = sc->tri_order[i];
        v3 l, h; tri_bounds(sc->d->tri_verts + 9 * (size_t)p, &l, &h);
        if (sc->tris[p].kind == 1) {             /* a rectangle is tested where its FIRST carrier lies: bound all four corners */
            v3 l2, h2; tri_bounds(sc->d->tri_verts + 9 * (size_t)(p + 1), &l2, &h2);
            if (l2.x < l.x) l.x = l2.x; if (l2.y < l.y) l.y = l2.y; if (l2.z < l.z) l.z = l2.z;
            if (h2.x > h.x) h.x = h2.x; if (h2.y > h.y) h.y = h2.y; if (h2.z > h.z) h.z = h2.z;
        }
        if (l.x < lo.x) lo.x = l.x; if (l.y < lo.y) lo.y = l.y; if (l.z < lo.z) lo.z = l.z;
        if (h.x > hi.x) hi.x = h.x; if (h.y > hi.y) hi.y = h.y; if (h.z > hi.z) hi.z = h.z;
    }
    /* pad: culling must be conservative, the triangle test alone decides hits */
    float ex = hi.x - lo.x, ey = hi.y - lo.y, ez = hi.z - lo.z;
    float pad = 1e-4f * (1.0f + fmaxf(ex, fmaxf(ey, ez)));
    N->lo = V(lo.x - pad, lo.y - pad, lo.z - pad); N->hi = V(hi.x + pad, hi.y + pad, hi.z + pad);
    N->first = first; N->count = count; N->left = N->right = -1;
    if (count <= 2) return idx;
    int axis = (ex >= ey && ex >= ez) ? 0 : (ey >= ez ? 1 : 2);
    g_sort_verts = sc->d->tri_verts; g_sort_axis = axis;
    qsort(sc->tri_order + first, (size_t)count, sizeof(int), cmp_centroid);
    int half = count / 2;
    int l = build_node(sc, first, half);
    int r = build_node(sc, first + half, count - half);
    sc->nodes[idx].left = l; sc->nodes[idx].right = r; sc->nodes[idx].count = 0;
    return idx;
}
static void build_bvh(orc_scene *sc)
{
    /* over every triangle; the leaf loop tests a rectangle when it meets its first carrier (kind 1, bounded by all
     * four corners in build_node) and skips the second (kind 2) */
    int n = (int)sc->d->n_tris;
    sc->tri_order = (int *)malloc(sizeof(int) * (size_t)(n ? n : 1));
    for (int i = 0; i < n; ++i) sc->tri_order[i] = i;
    sc->nodes = (orc_node *)malloc(sizeof(orc_node) * (2u * (size_t)(n > 0 ? n : 0) + 1u));
    sc->n_nodes = 0;
    if (n) build_node(sc, 0, n);
}
static void free_scene(orc_scene *sc)
{
    free(sc->tris); free(sc->em_n); free(sc->em_inv_area); free(sc->em_face_pmf); free(sc->em_face_cdf); free(sc->nodes); free(sc->tri_order);
}

/* ------------------------------------------------------------------ */
/* Ray / triangle  [mitsuba3: Mesh::ray_intersect_triangle_impl]       */
/* ------------------------------------------------------------------ */
typedef struct { v3 o, d; float maxt; } ray3;
typedef struct { float t, u, v; int prim; } hit_t;

/* Shared edges are CLOSED: a barycentric coordinate may undershoot its edge by MTR_EDGE_EPS (2^-19 of the triangle).
 * Moller-Trumbore evaluates the two triangles of an edge with different operation orders, so a ray aimed exactly at the
 * edge can fail both tests by one rounding.  That is no measure-zero event when a render connects millions of paths to ONE
 * point: the laser spot of examples/transient-nlos/nlos-z-simple.xml is the centre of a two-triangle wall, i.e. ON its
 * diagonal, and 11 % of the connections of emitter_laser_targets_sample (transientnlospath.py:511-564) fell through
 * (frames 0.886 of the reference's notebook figure; with closed edges 0.985 ... 1.0).  The reference
 * traces with Embree, whose Moller-Trumbore variant evaluates an edge at the triangles' common first vertex with the same
 * expression for both (U = dot(cross(v0 - o, d), e2) of one is -V of the other: no gap on a quad's diagonal).  A ray on
 * the edge now hits BOTH triangles at the same t; the tie goes to the lower original index as always. */
#define MTR_EDGE_EPS 1.9073486328125e-06f
static inline int tri_test(const orc_tri *T, const ray3 *r, float *t, float *u, float *v)
{
    v3 pvec = vcross(r->d, T->e2);
    float det = vdot(T->e1, pvec);
    float inv_det = 1.0f / det;
    v3 tvec = vsub(r->o, T->p0);
    float uu = vdot(tvec, pvec) * inv_det;
    if (!(uu >= -MTR_EDGE_EPS)) return 0;
    v3 qvec = vcross(tvec, T->e1);
    float vv = vdot(r->d, qvec) * inv_det;
    float ww = 1.0f - (uu + vv);                  /* the third barycentric coordinate; mitsuba's u <= 1 follows from the three */
    if (!(vv >= -MTR_EDGE_EPS && ww >= -MTR_EDGE_EPS)) return 0;
    float tt = vdot(T->e2, qvec) * inv_det;
    if (!(tt >= 0.0f && tt <= r->maxt)) return 0;
    *t = tt; *u = uu; *v = vv;
    return 1;
}
/* [mitsuba3: Rectangle::ray_intersect_preliminary_impl]
 *   ray = to_object.transform_affine(ray_);  t = -ray.o.z / ray.d.z;  local = ray(t);
 *   active = t >= 0 && t <= maxt && |local.x| <= 1 && |local.y| <= 1;   prim_uv = (local.x, local.y)
 * transform_affine(point): result = translation column, then fmadd(column_i, p[i], result) for i = 0, 1, 2;
 * (vector): column_0 * v[0], then fmadd for i = 1, 2. */
static inline float xf_point(const float r[4], v3 p) { return fmaf(r[2], p.z, fmaf(r[1], p.y, fmaf(r[0], p.x, r[3]))); }
static inline float xf_vec(const float r[4], v3 v) { return fmaf(r[2], v.z, fmaf(r[1], v.y, r[0] * v.x)); }
static inline int quad_test(const orc_tri *Q, const ray3 *r, float *t, float *u, float *v)
{
    float oz = xf_point(Q->rz, r->o), dz = xf_vec(Q->rz, r->d);
    float tt = -oz / dz;
    float lx = fmaf(xf_vec(Q->rx, r->d), tt, xf_point(Q->rx, r->o));       /* ray(t) = fmadd(d, t, o) */
    float ly = fmaf(xf_vec(Q->ry, r->d), tt, xf_point(Q->ry, r->o));
    if (!(tt >= 0.0f && tt <= r->maxt && fabsf(lx) <= 1.0f && fabsf(ly) <= 1.0f)) return 0;
    *t = tt; *u = lx; *v = ly;
    return 1;
}
/* one primitive: mesh triangle, rectangle (at its first carrier) or nothing (second carrier) */
static inline int prim_test(const orc_tri *T, const ray3 *r, float *t, float *u, float *v)
{
    if (T->kind == 0) return tri_test(T, r, t, u, v);
    if (T->kind == 1) return quad_test(T, r, t, u, v);
    return 0;
}
/* closest hit; ties on t are broken towards the LOWER primitive index so the
 * result does not depend on traversal order (brute force == any BVH) */
static inline void hit_update(hit_t *h, float t, float u, float v, int prim)
{
    if (t < h->t || (t == h->t && prim < h->prim)) { h->t = t; h->u = u; h->v = v; h->prim = prim; }
}
/* culling only (conservative: padded boxes); plain ternaries instead of libm fminf/fmaxf calls */
#define ORC_MIN(a, b) ((a) < (b) ? (a) : (b))
#define ORC_MAX(a, b) ((a) > (b) ? (a) : (b))
static inline int box_hit(const orc_node *N, const ray3 *r, v3 inv_d, float tbest)
{
    float t0 = (N->lo.x - r->o.x) * inv_d.x, t1 = (N->hi.x - r->o.x) * inv_d.x;
    float tn = ORC_MIN(t0, t1), tf = ORC_MAX(t0, t1);
    t0 = (N->lo.y - r->o.y) * inv_d.y; t1 = (N->hi.y - r->o.y) * inv_d.y;
    float a = ORC_MIN(t0, t1), b = ORC_MAX(t0, t1);
    tn = ORC_MAX(tn, a); tf = ORC_MIN(tf, b);
    t0 = (N->lo.z - r->o.z) * inv_d.z; t1 = (N->hi.z - r->o.z) * inv_d.z;
    a = ORC_MIN(t0, t1); b = ORC_MAX(t0, t1);
    tn = ORC_MAX(tn, a); tf = ORC_MIN(tf, b);
    tf *= 1.0000005f;
    /* written so that a NaN (0 * inf on an axis-parallel ray) never culls */
    return !(tn > tf) && !(tf < 0.0f) && !(tn > tbest);
}
static hit_t intersect(const orc_scene *sc, const ray3 *r, int use_bvh)
{
    hit_t h; h.t = INFINITY; h.u = h.v = 0.0f; h.prim = -1;
    float t, u, v;
    if (!use_bvh) {
        for (uint32_t i = 0; i < sc->d->n_tris; ++i)
            if (prim_test(&sc->tris[i], r, &t, &u, &v)) hit_update(&h, t, u, v, (int)i);
        return h;
    }
    if (!sc->n_nodes) return h;
    v3 inv_d = V(1.0f / r->d.x, 1.0f / r->d.y, 1.0f / r->d.z);
    int stack[128], sp = 0; stack[sp++] = 0;
    while (sp) {
        const orc_node *N = &sc->nodes[stack[--sp]];
        /* (x (1 + 2^-10): the computed distance of a grazing sliver can lie in front of its own box — the product's kCullSlack, mtr_core.h) */
        if (!box_hit(N, r, inv_d, ORC_MIN(h.t, r->maxt) * 1.0009765625f)) continue;
        if (N->left < 0) {
            for (int i = N->first; i < N->first + N->count; ++i) {
                int p = sc->tri_order[i];
                if (prim_test(&sc->tris[p], r, &t, &u, &v)) hit_update(&h, t, u, v, p);
            }
        } else { stack[sp++] = N->left; stack[sp++] = N->right; }
    }
    return h;
}
/* [mitsuba3: Scene::ray_test] any hit in [0, maxt] */
static int ray_test(const orc_scene *sc, const ray3 *r, int use_bvh)
{
    float t, u, v;
    if (!use_bvh) {
        for (uint32_t i = 0; i < sc->d->n_tris; ++i)
            if (prim_test(&sc->tris[i], r, &t, &u, &v)) return 1;
        return 0;
    }
    if (!sc->n_nodes) return 0;
    v3 inv_d = V(1.0f / r->d.x, 1.0f / r->d.y, 1.0f / r->d.z);
    int stack[128], sp = 0; stack[sp++] = 0;
    while (sp) {
        const orc_node *N = &sc->nodes[stack[--sp]];
        if (!box_hit(N, r, inv_d, r->maxt * 1.0009765625f)) continue;
        if (N->left < 0) {
            for (int i = N->first; i < N->first + N->count; ++i)
                if (prim_test(&sc->tris[sc->tri_order[i]], r, &t, &u, &v)) return 1;
        } else { stack[sp++] = N->left; stack[sp++] = N->right; }
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* Surface interaction  [mitsuba3: Mesh::compute_surface_interaction]  */
/* ------------------------------------------------------------------ */
typedef struct {
    int valid; float t; v3 p, n, s, tt, wi; int prim;       /* n, s, tt: the shading frame (si.sh_frame) */
    float uv[2];                                               /* si.uv */
    v3 ng;                                                     /* si.n: the geometric normal (ray offsets, hidden_geometry_sample's cos_theta_i) */
} sinter;

static sinter make_si(const orc_scene *sc, const ray3 *r, hit_t h)
{
    sinter si; memset(&si, 0, sizeof si);
    si.valid = h.prim >= 0; si.t = h.t; si.prim = h.prim;
    if (!si.valid) return si;
    const orc_tri *T = &sc->tris[h.prim];
    const float *vv = sc->d->tri_verts + 9 * (size_t)h.prim;
    float b1 = h.u, b2 = h.v, b0 = 1.0f - b1 - b2;
    /* si.p = fmadd(p0, b0, fmadd(p1, b1, p2 * b2)) */
    si.p = V(fmaf(vv[0], b0, fmaf(vv[3], b1, vv[6] * b2)),
             fmaf(vv[1], b0, fmaf(vv[4], b1, vv[7] * b2)),
             fmaf(vv[2], b0, fmaf(vv[5], b1, vv[8] * b2)));
    if (T->kind == 1)        /* [Rectangle::compute_surface_interaction] si.p = to_world.transform_affine((prim_uv.x, prim_uv.y, 0)) */
        si.p = V(fmaf(T->qdv.x, h.v, fmaf(T->qdu.x, h.u, T->qc.x)),
                 fmaf(T->qdv.y, h.v, fmaf(T->qdu.y, h.u, T->qc.y)),
                 fmaf(T->qdv.z, h.v, fmaf(T->qdu.z, h.u, T->qc.z)));
    si.n = T->n; si.s = T->s; si.tt = T->t; si.ng = T->n;
    /* [Mesh::compute_surface_interaction] si.uv = fmadd(uv2, b2, fmadd(uv1, b1, uv0 * b0)); [Rectangle] (prim_uv + 1) / 2 */
    if (T->kind == 1) { si.uv[0] = fmaf(h.u, 0.5f, 0.5f); si.uv[1] = fmaf(h.v, 0.5f, 0.5f); }
    else {
        /* without vertex texture coordinates (none, or all zero for this shape's triangles): si.uv = (b1, b2) */
        static const float bary[6] = { 0.0f, 0.0f, 1.0f, 0.0f, 0.0f, 1.0f };
        const float *uv = sc->d->tri_uv ? sc->d->tri_uv + 6 * (size_t)h.prim : bary;
        int any_uv = 0;
        for (int k = 0; k < 6; ++k) any_uv |= uv[k] != 0.0f;
        if (!any_uv) uv = bary;
        si.uv[0] = fmaf(uv[4], b2, fmaf(uv[2], b1, uv[0] * b0)); si.uv[1] = fmaf(uv[5], b2, fmaf(uv[3], b1, uv[1] * b0));
    }
    if (T->smooth) {        /* sh_frame.n = normalize(fmadd(n2, b2, fmadd(n1, b1, n0 * b0))), then initialize_sh_frame */
        const float *vn = sc->d->tri_normals + 9 * (size_t)h.prim;
        si.n = vnormalize(V(fmaf(vn[6], b2, fmaf(vn[3], b1, vn[0] * b0)), fmaf(vn[7], b2, fmaf(vn[4], b1, vn[1] * b0)),
                            fmaf(vn[8], b2, fmaf(vn[5], b1, vn[2] * b0))));
        sh_frame_from(si.n, T->dp_du, &si.s, &si.tt);
    }
    v3 md = vneg(r->d);
    si.wi = V(vdot(md, si.s), vdot(md, si.tt), vdot(md, si.n));   /* to_local(-ray.d) */
    return si;
}
/* [mitsuba3: BitmapTexture::eval, filter_type = bilinear, wrap_mode = repeat] (restated; upstream not under /root/reference):
 * uv -> texel space (u w - 1/2, v h - 1/2), four neighbours wrapped by the positive modulo,
 * fmadd(w0.y, fmadd(w0.x, v00, w1.x v10), w1.y fmadd(w0.x, v01, w1.x v11)) */
static void texture_eval(const mtr_texture *T, float u, float v, float out[3])
{
    float fu = fmaf(u, (float)T->width, -0.5f), fv = fmaf(v, (float)T->height, -0.5f);
    float flu = floorf(fu), flv = floorf(fv);
    float w1x = fu - flu, w1y = fv - flv, w0x = 1.0f - w1x, w0y = 1.0f - w1y;
    int32_t ix = (int32_t)flu, iy = (int32_t)flv, W = (int32_t)T->width, H = (int32_t)T->height;
    int32_t x0 = ix % W, x1 = (ix + 1) % W, y0 = iy % H, y1 = (iy + 1) % H;
    x0 += x0 < 0 ? W : 0; x1 += x1 < 0 ? W : 0; y0 += y0 < 0 ? H : 0; y1 += y1 < 0 ? H : 0;
    const float *v00 = T->rgb + 3 * ((size_t)y0 * W + x0), *v10 = T->rgb + 3 * ((size_t)y0 * W + x1);
    const float *v01 = T->rgb + 3 * ((size_t)y1 * W + x0), *v11 = T->rgb + 3 * ((size_t)y1 * W + x1);
    for (int k = 0; k < 3; ++k) {
        float f0 = fmaf(w0x, v00[k], w1x * v10[k]), f1 = fmaf(w0x, v01[k], w1x * v11[k]);
        out[k] = fmaf(w0y, f0, w1y * f1);
    }
}
void orc_texture_eval(const mtr_texture *T, uint32_t n, const float *u, const float *v, float *out3)
{
    for (uint32_t i = 0; i < n; ++i) texture_eval(T, u[i], v[i], out3 + 3 * i);
}
/* a material with its colour `a` taken from its bitmap at the hit (a copy), or the material itself */
static const mtr_material *material_at(const mtr_scene_desc *d, const mtr_material *m, const sinter *si, mtr_material *copy)
{
    if (!m || m->albedo_texture == 0u || !d->textures || m->albedo_texture > d->n_textures) return m;
    *copy = *m;
    texture_eval(&d->textures[m->albedo_texture - 1u], si->uv[0], si->uv[1], copy->a);
    return copy;
}
/* [mitsuba3: Frame3f::to_world] fmadd(n, v.z, fmadd(t, v.y, s * v.x)) */
static v3 to_world(const sinter *si, v3 v)
{
    return V(fmaf(si->n.x, v.z, fmaf(si->tt.x, v.y, si->s.x * v.x)),
             fmaf(si->n.y, v.z, fmaf(si->tt.y, v.y, si->s.y * v.x)),
             fmaf(si->n.z, v.z, fmaf(si->tt.z, v.y, si->s.z * v.x)));
}
static v3 to_local(const sinter *si, v3 v) { return V(vdot(v, si->s), vdot(v, si->tt), vdot(v, si->n)); }
/* [mitsuba3: Interaction::offset_p] */
static v3 offset_p(const sinter *si, v3 d)
{
    float m = fmaxf(fabsf(si->p.x), fmaxf(fabsf(si->p.y), fabsf(si->p.z)));
    float mag = (1.0f + m) * ORC_RAY_EPS;
    mag = mulsign(mag, vdot(si->ng, d));
    return vfma(si->ng, mag, si->p);
}

/* ------------------------------------------------------------------ */
/* BSDFs  [mitsuba3: src/bsdfs/{diffuse,conductor,dielectric,twosided}.cpp,
 *         include/mitsuba/render/fresnel.h]                           */
/* ------------------------------------------------------------------ */
typedef struct { v3 wo; float pdf, eta; int delta; float w[3]; } bsample;

static float fresnel_conductor(float cos_i, float eta_r, float eta_i)
{
    float c2 = cos_i * cos_i, s2 = 1.0f - c2, s4 = s2 * s2;
    float temp1 = eta_r * eta_r - eta_i * eta_i - s2;
    float q = temp1 * temp1 + 4.0f * eta_i * eta_i * eta_r * eta_r;
    float a2pb2 = sqrtf(q > 0.0f ? q : 0.0f);
    float h = 0.5f * (a2pb2 + temp1);
    float a = sqrtf(h > 0.0f ? h : 0.0f);
    float term1 = a2pb2 + c2, term2 = 2.0f * cos_i * a;
    float rs = (term1 - term2) / (term1 + term2);
    float term3 = a2pb2 * c2 + s4, term4 = term2 * s2;
    float rp = rs * (term3 - term4) / (term3 + term4);
    return 0.5f * (rs + rp);
}
static void fresnel_dielectric(float cos_i, float eta, float *r, float *cos_t, float *eta_it, float *eta_ti)
{
    int outside = cos_i >= 0.0f;
    float rcp_eta = 1.0f / eta;
    *eta_it = outside ? eta : rcp_eta; *eta_ti = outside ? rcp_eta : eta;
    float ct2 = fmaf(-fmaf(-cos_i, cos_i, 1.0f), (*eta_ti) * (*eta_ti), 1.0f);
    float ci = fabsf(cos_i), ct = sqrtf(ct2 > 0.0f ? ct2 : 0.0f);
    int index_matched = (eta == 1.0f), special = index_matched || (ci == 0.0f);
    float a_s = fmaf(-(*eta_it), ct, ci) / fmaf(*eta_it, ct, ci);
    float a_p = fmaf(-(*eta_it), ci, ct) / fmaf(*eta_it, ci, ct);
    float rr = 0.5f * (a_s * a_s + a_p * a_p);
    if (special) rr = index_matched ? 0.0f : 1.0f;
    *r = rr;
    *cos_t = (cos_i < 0.0f || (cos_i == 0.0f && signbit(cos_i))) ? ct : -ct;   /* mulsign_neg */
}
/* ---- GGX microfacet lobes: mitsuba 3's MicrofacetDistribution (isotropic alpha, sample_visible = true), RoughConductor and
 * RoughPlastic, restated from the published source (upstream not present under /root/reference: unverified here).
 * [MicrofacetDistribution::eval]: 1 / (pi alpha_u alpha_v (sqr(m.x/alpha_u) + sqr(m.y/alpha_v) + sqr(m.z))^2), 0 when D cos <= 1e-20 */
static float ggx_eval(v3 m, float au, float av)
{
    float mx = m.x / au, my = m.y / av;
    float t = fmaf(m.z, m.z, fmaf(my, my, mx * mx));
    float result = 1.0f / (((ORC_PI * (au * av)) * t) * t);
    return (result * m.z > 1e-20f) ? result : 0.0f;
}
/* [MicrofacetDistribution::smith_g1] */
static float ggx_smith_g1(v3 v, v3 m, float au, float av)
{
    float ax = au * v.x, ay = av * v.y;
    float xy_alpha_2 = fmaf(ay, ay, ax * ax);
    float tan_theta_alpha_2 = xy_alpha_2 / (v.z * v.z);
    float result = 2.0f / (1.0f + sqrtf(1.0f + tan_theta_alpha_2));
    if (xy_alpha_2 == 0.0f) result = 1.0f;                 /* perpendicular incidence: no shadowing / masking */
    if (vdot(v, m) * v.z <= 0.0f) result = 0.0f;           /* the back of the microfacet is not seen from the front */
    return result;
}
/* [MicrofacetDistribution::sample_visible_11] (GGX) */
static void ggx_sample_visible_11(float cos_theta_i, float u1, float u2, float *sx, float *sy)
{
    float px, py;
    square_to_disk(u1, u2, &px, &py);
    float s = 0.5f * (1.0f + cos_theta_i);
    float a0 = fmaf(-px, px, 1.0f);
    float a = sqrtf(a0 > 0.0f ? a0 : 0.0f);
    py = fmaf(py, s, fmaf(-a, s, a));                      /* lerp(safe_sqrt(1 - p.x^2), p.y, s) */
    float z0 = 1.0f - fmaf(py, py, px * px);
    float z = sqrtf(z0 > 0.0f ? z0 : 0.0f);
    float si0 = fmaf(-cos_theta_i, cos_theta_i, 1.0f);
    float sin_theta_i = sqrtf(si0 > 0.0f ? si0 : 0.0f);
    float norm = 1.0f / fmaf(sin_theta_i, py, cos_theta_i * z);
    *sx = fmaf(cos_theta_i, py, -(sin_theta_i * z)) * norm;
    *sy = px * norm;
}
/* ---- Beckmann lobes (MTR_MAT_BECKMANN; mitsuba's default `distribution`), restated from the published MicrofacetDistribution.
 * exp / log / erf / erfinv are restated with explicit operation order, the same sequences as the product's mtr_core.h (libm, ocml
 * and drjit each round them differently): Cephes-style expf / logf, Abramowitz & Stegun 7.1.28 for erf, M. Giles' single-precision
 * erfinv.  tests/test_rough_bsdf.py::test_special_functions holds them to scipy. */
static float orc_bitsf(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static uint32_t orc_fbits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static float orc_expf(float x)
{
    if (!(x > -87.0f)) return 0.0f;
    if (x > 88.0f) x = 88.0f;
    float n = floorf(fmaf(x, 1.44269504088896341f, 0.5f));
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    float y = fmaf(p, r * r, r) + 1.0f;
    return y * orc_bitsf((uint32_t)((int32_t)n + 127) << 23);
}
static float orc_logf(float x)
{
    uint32_t b = orc_fbits(x);
    int32_t e = (int32_t)(b >> 23) - 127;
    float m = orc_bitsf((b & 0x007fffffu) | 0x3f800000u);
    if (m > 1.41421356237f) { m *= 0.5f; e += 1; }
    float f = m - 1.0f, z = f * f, fe = (float)e;
    float p = 7.0376836292e-2f;
    p = fmaf(p, f, -1.1514610310e-1f);
    p = fmaf(p, f, 1.1676998740e-1f);
    p = fmaf(p, f, -1.2420140846e-1f);
    p = fmaf(p, f, 1.4249322787e-1f);
    p = fmaf(p, f, -1.6668057665e-1f);
    p = fmaf(p, f, 2.0000714765e-1f);
    p = fmaf(p, f, -2.4999993993e-1f);
    p = fmaf(p, f, 3.3333331174e-1f);
    float y = (p * f) * z;
    y = fmaf(fe, -2.12194440e-4f, y);
    y = fmaf(-0.5f, z, y);
    return fmaf(fe, 0.693359375f, f + y);
}
static float orc_erff(float x)
{
    float a = fabsf(x);
    float t = 0.0000430638f;
    t = fmaf(t, a, 0.0002765672f);
    t = fmaf(t, a, 0.0001520143f);
    t = fmaf(t, a, 0.0092705272f);
    t = fmaf(t, a, 0.0422820123f);
    t = fmaf(t, a, 0.0705230784f);
    t = fmaf(t, a, 1.0f);
    t = t * t; t = t * t; t = t * t; t = t * t;
    float r = 1.0f - 1.0f / t;
    return x < 0.0f ? -r : r;
}
static float orc_erfinvf(float x)
{
    float w = -orc_logf((1.0f - x) * (1.0f + x));
    float p;
    if (w < 5.0f) {
        w = w - 2.5f;
        p = 2.81022636e-08f;
        p = fmaf(p, w, 3.43273939e-07f);
        p = fmaf(p, w, -3.5233877e-06f);
        p = fmaf(p, w, -4.39150654e-06f);
        p = fmaf(p, w, 0.00021858087f);
        p = fmaf(p, w, -0.00125372503f);
        p = fmaf(p, w, -0.00417768164f);
        p = fmaf(p, w, 0.246640727f);
        p = fmaf(p, w, 1.50140941f);
    } else {
        w = sqrtf(w) - 3.0f;
        p = -0.000200214257f;
        p = fmaf(p, w, 0.000100950558f);
        p = fmaf(p, w, 0.00134934322f);
        p = fmaf(p, w, -0.00367342844f);
        p = fmaf(p, w, 0.00573950773f);
        p = fmaf(p, w, -0.0076224613f);
        p = fmaf(p, w, 0.00943887047f);
        p = fmaf(p, w, 1.00167406f);
        p = fmaf(p, w, 2.83297682f);
    }
    return p * x;
}
/* test hook: which = 0 exp, 1 log, 2 erf, 3 erfinv */
void orc_special(int which, uint64_t n, const float *x, float *y)
{
    for (uint64_t i = 0; i < n; ++i)
        y[i] = which == 0 ? orc_expf(x[i]) : which == 1 ? orc_logf(x[i]) : which == 2 ? orc_erff(x[i]) : orc_erfinvf(x[i]);
}
/* [MicrofacetDistribution::eval] (Beckmann): exp(-(sqr(m.x/alpha_u) + sqr(m.y/alpha_v)) / cos^2) / (pi alpha_u alpha_v cos^4) */
static float beck_eval(v3 m, float au, float av)
{
    float mx = m.x / au, my = m.y / av;
    float c2 = m.z * m.z;
    float result = orc_expf(-fmaf(my, my, mx * mx) / c2) / ((ORC_PI * (au * av)) * (c2 * c2));
    return (result * m.z > 1e-20f) ? result : 0.0f;
}
/* [MicrofacetDistribution::smith_g1] (Beckmann): a = rsqrt(tan_theta_alpha_2); 1 for a >= 1.6, else the rational approximation */
static float beck_smith_g1(v3 v, v3 m, float au, float av)
{
    float ax = au * v.x, ay = av * v.y;
    float xy_alpha_2 = fmaf(ay, ay, ax * ax);
    float tan_theta_alpha_2 = xy_alpha_2 / (v.z * v.z);
    float a = 1.0f / sqrtf(tan_theta_alpha_2), a_sqr = a * a;
    float result = (a >= 1.6f) ? 1.0f : fmaf(2.181f, a_sqr, 3.535f * a) / fmaf(2.577f, a_sqr, fmaf(2.276f, a, 1.0f));
    if (xy_alpha_2 == 0.0f) result = 1.0f;
    if (vdot(v, m) * v.z <= 0.0f) result = 0.0f;
    return result;
}
/* [MicrofacetDistribution::sample_visible_11] (Beckmann): numerical inversion in the erf domain — first guess, three Newton
 * iterations; x is clamped inside (-1, 1) before every erfinv (an overshooting step must not yield a NaN slope) */
static void beck_sample_visible_11(float cos_theta_i, float u1, float u2, float *sx, float *sy)
{
    const float kInvSqrtPi = 0.56418958354775628695f, kEdge = 0.999999f;
    float s0 = fmaf(-cos_theta_i, cos_theta_i, 1.0f);
    float tan_theta_i = sqrtf(s0 > 0.0f ? s0 : 0.0f) / cos_theta_i;
    float cot_theta_i = 1.0f / tan_theta_i;
    float maxval = orc_erff(cot_theta_i);
    u1 = fmaxf(fminf(u1, 1.0f - 1e-6f), 1e-6f); u2 = fmaxf(fminf(u2, 1.0f - 1e-6f), 1e-6f);
    float x = maxval - (maxval + 1.0f) * orc_erff(sqrtf(-orc_logf(u1)));
    float target = u1 * ((1.0f + maxval) + (kInvSqrtPi * tan_theta_i) * orc_expf(-(cot_theta_i * cot_theta_i)));
    for (int it = 0; it < 3; ++it) {
        x = fmaxf(fminf(x, kEdge), -kEdge);
        float slope = orc_erfinvf(x);
        float value = fmaf(kInvSqrtPi * tan_theta_i, orc_expf(-(slope * slope)), 1.0f + x) - target;
        float derivative = 1.0f - slope * tan_theta_i;
        x -= value / derivative;
    }
    x = fmaxf(fminf(x, kEdge), -kEdge);
    *sx = orc_erfinvf(x);
    *sy = orc_erfinvf(fmaf(2.0f, u2, -1.0f));
}
static float mf_eval(v3 m, float au, float av, int beck) { return beck ? beck_eval(m, au, av) : ggx_eval(m, au, av); }
static float mf_smith_g1(v3 v, v3 m, float au, float av, int beck) { return beck ? beck_smith_g1(v, m, au, av) : ggx_smith_g1(v, m, au, av); }
/* [RoughConductor: alpha_u, alpha_v] MTR_MAT_ANISOTROPIC: the roughness along the bitangent travels in c2[0] */
static float rough_alpha_v(const mtr_material *m)
{
    return (m->flags & MTR_MAT_ANISOTROPIC) ? (m->type == MTR_BSDF_ROUGHDIELECTRIC ? m->b[0] : m->c2[0]) : m->alpha;
}

/* [MicrofacetDistribution::sample], visible normals: stretch, sample the slope, rotate + unstretch, normal and density */
static v3 ggx_sample(v3 wi, float au, float av, float u1, float u2, float *pdf, int beck)
{
    v3 wi_p = vnormalize(V(au * wi.x, av * wi.y, wi.z));
    float sin_theta_2 = fmaf(-wi_p.z, wi_p.z, 1.0f);
    float sin_phi = 0.0f, cos_phi = 1.0f;                  /* [Frame3f::sincos_phi] */
    if (fabsf(sin_theta_2) > 4.0f * 5.9604644775390625e-8f) {
        float inv = 1.0f / sqrtf(sin_theta_2);
        sin_phi = fminf(fmaxf(wi_p.y * inv, -1.0f), 1.0f); cos_phi = fminf(fmaxf(wi_p.x * inv, -1.0f), 1.0f);
    }
    float sx, sy;
    if (beck) beck_sample_visible_11(wi_p.z, u1, u2, &sx, &sy);
    else ggx_sample_visible_11(wi_p.z, u1, u2, &sx, &sy);
    float rx = fmaf(cos_phi, sx, -(sin_phi * sy)) * au;
    float ry = fmaf(sin_phi, sx, cos_phi * sy) * av;
    v3 m = vnormalize(V(-rx, -ry, 1.0f));
    *pdf = ((mf_eval(m, au, av, beck) * mf_smith_g1(wi, m, au, av, beck)) * fabsf(vdot(wi, m))) / wi.z;
    return m;
}
/* [RoughPlastic: dr::lerp_gather over m_external_transmittance, MI_ROUGH_TRANSMITTANCE_RES = 64] */
static float rough_transmittance(const mtr_material *m, float cos_theta)
{
    float x = cos_theta * (float)(MTR_ROUGH_TRANSMITTANCE_RES - 1);
    uint32_t i = (uint32_t)x;
    if (i > MTR_ROUGH_TRANSMITTANCE_RES - 2u) i = MTR_ROUGH_TRANSMITTANCE_RES - 2u;
    float w1 = x - (float)i, w0 = 1.0f - w1;
    return fmaf(w0, m->external_transmittance[i], w1 * m->external_transmittance[i + 1u]);
}
static int bsdf_is_rough(const mtr_material *m)          /* the smooth lobes beyond `diffuse`: microfacet lobes and plastic's base */
{
    return m->type == MTR_BSDF_ROUGHCONDUCTOR || m->type == MTR_BSDF_ROUGHPLASTIC || m->type == MTR_BSDF_ROUGHDIELECTRIC || m->type == MTR_BSDF_PLASTIC;
}
/* [mitsuba3: src/bsdfs/plastic.cpp — Plastic::eval / ::pdf; restated from the published source].  The coat is a delta lobe: eval and
 * pdf see the diffuse base only — diff / (1 - (nonlinear ? diff fdr_int : fdr_int)) * cos/pi * 1/eta^2 * (1 - F_i)(1 - F_o); the lobe
 * probabilities are F_i ssw and (1 - F_i)(1 - ssw), normalised.  internal_reflectance = m_fdr_int, computed by the caller. */
static void plastic_probs(const mtr_material *m, float f_i, float *ps, float *pdif)
{
    *ps = f_i * m->specular_sampling_weight; *pdif = (1.0f - f_i) * (1.0f - m->specular_sampling_weight);
    *ps = *ps / (*ps + *pdif); *pdif = 1.0f - *ps;
}
static void plastic_eval_pdf(const mtr_material *m, v3 wi, v3 wo, float val[3], float *pdf)
{
    val[0] = val[1] = val[2] = 0.0f; *pdf = 0.0f;
    float ci = wi.z, co = wo.z;
    if (!(ci > 0.0f && co > 0.0f)) return;
    float eta = m->int_ior / m->ext_ior, inv_eta_2 = 1.0f / (eta * eta);
    float f_i, f_o, ct, eit, eti;
    fresnel_dielectric(ci, eta, &f_i, &ct, &eit, &eti);
    fresnel_dielectric(co, eta, &f_o, &ct, &eit, &eti);
    float ps, pdif;
    plastic_probs(m, f_i, &ps, &pdif);
    float cpdf = ORC_INV_PI * co;
    *pdf = cpdf * pdif;
    float scale = ((cpdf * inv_eta_2) * (1.0f - f_i)) * (1.0f - f_o);
    for (int k = 0; k < 3; ++k)
        val[k] = (m->a[k] / (1.0f - ((m->flags & MTR_MAT_NONLINEAR) ? m->a[k] * m->internal_reflectance : m->internal_reflectance))) * scale;
}
static float fresnel_conductor(float cos_i, float eta_r, float eta_i);
static void fresnel_dielectric(float cos_i, float eta, float *r, float *cos_t, float *eta_it, float *eta_ti);
/* [RoughConductor::eval / ::pdf], [RoughPlastic::eval / ::pdf]; wi, wo local, already on the two-sided side */
/* [mitsuba3: src/bsdfs/roughdielectric.cpp — RoughDielectric::eval / ::pdf, sample_visible = true, TransportMode::Radiance; restated
 * from the published source, unverified here].  reflect = cos_theta_i cos_theta_o > 0; m = normalize(wi + wo * (reflect ? 1 : eta)),
 * flipped into the macro normal's hemisphere; F D G / (4 |cos_i|) or |scale (1 - F) D G eta^2 (wi.m)(wo.m) / (cos_i (wi.m + eta wo.m)^2)|,
 * scale = 1 / eta^2; pdf = distr.pdf(mulsign(wi, cos_i), m) |dwh_dwo| (F | 1 - F) where micro- and macro-surface agree on the sides */
static v3 mulsign3(v3 v, float s) { return (s < 0.0f || (s == 0.0f && signbit(s))) ? V(-v.x, -v.y, -v.z) : v; }
static void rough_dielectric_eval_pdf(const mtr_material *m, v3 wi, v3 wo, float val[3], float *pdf)
{
    val[0] = val[1] = val[2] = 0.0f; *pdf = 0.0f;
    float ci = wi.z, co = wo.z;
    if (ci == 0.0f) return;
    int beck = (m->flags & MTR_MAT_BECKMANN) != 0u;
    float au = m->alpha, av = rough_alpha_v(m);
    int reflect = ci * co > 0.0f;
    float eta_m = m->int_ior / m->ext_ior, inv_eta_m = m->ext_ior / m->int_ior;
    float eta = ci > 0.0f ? eta_m : inv_eta_m, inv_eta = ci > 0.0f ? inv_eta_m : eta_m;
    float sc = reflect ? 1.0f : eta;
    v3 h = vnormalize(V(fmaf(wo.x, sc, wi.x), fmaf(wo.y, sc, wi.y), fmaf(wo.z, sc, wi.z)));
    h = mulsign3(h, h.z);
    float D = mf_eval(h, au, av, beck);
    float wih = vdot(wi, h), woh = vdot(wo, h);
    float F, ct, eit, eti;
    fresnel_dielectric(wih, eta_m, &F, &ct, &eit, &eti);
    float g1i = mf_smith_g1(wi, h, au, av, beck);
    float G = g1i * mf_smith_g1(wo, h, au, av, beck);
    float t = fmaf(eta, woh, wih);
    if (reflect) {
        float v = ((F * D) * G) / (4.0f * fabsf(ci));
        for (int k = 0; k < 3; ++k) val[k] = m->c[k] * v;
    } else {
        float scale = inv_eta * inv_eta;
        float v = fabsf(((((((scale * (1.0f - F)) * D) * G) * eta) * eta) * wih) * woh / (ci * (t * t)));
        for (int k = 0; k < 3; ++k) val[k] = m->c2[k] * v;
    }
    if (!(wih * ci > 0.0f && woh * co > 0.0f)) return;
    float dwh_dwo = reflect ? 1.0f / (4.0f * woh) : ((eta * eta) * woh) / (t * t);
    v3 wu = mulsign3(wi, ci);
    float pm = ((D * mf_smith_g1(wu, h, au, av, beck)) * fabsf(vdot(wu, h))) / wu.z;
    *pdf = fabsf((pm * dwh_dwo) * (reflect ? F : 1.0f - F));
}
static void rough_eval_pdf(const mtr_material *m, v3 wi, v3 wo, float val[3], float *pdf)
{
    if (m->type == MTR_BSDF_ROUGHDIELECTRIC) { rough_dielectric_eval_pdf(m, wi, wo, val, pdf); return; }
    if (m->type == MTR_BSDF_PLASTIC) { plastic_eval_pdf(m, wi, wo, val, pdf); return; }
    val[0] = val[1] = val[2] = 0.0f; *pdf = 0.0f;
    float ci = wi.z, co = wo.z;
    if (!(ci > 0.0f && co > 0.0f)) return;
    v3 H = vnormalize(V(wo.x + wi.x, wo.y + wi.y, wo.z + wi.z));
    int beck = (m->flags & MTR_MAT_BECKMANN) != 0u;
    float au = m->alpha, av = rough_alpha_v(m);
    float D = mf_eval(H, au, av, beck);
    float g1i = mf_smith_g1(wi, H, au, av, beck);
    if (m->type == MTR_BSDF_ROUGHCONDUCTOR) {
        float wih = vdot(wi, H);
        if (wih > 0.0f && vdot(wo, H) > 0.0f) *pdf = (D * g1i) / (4.0f * ci);
        if (D != 0.0f) {
            float G = g1i * mf_smith_g1(wo, H, au, av, beck);
            float r = (D * G) / (4.0f * ci);
            for (int k = 0; k < 3; ++k) val[k] = (r * fresnel_conductor(wih, m->a[k], m->b[k])) * m->c[k];
        }
        return;
    }
    float t_i = rough_transmittance(m, ci), t_o = rough_transmittance(m, co);
    float ps = (1.0f - t_i) * m->specular_sampling_weight, pdif = t_i * (1.0f - m->specular_sampling_weight);
    ps = ps / (ps + pdif); pdif = 1.0f - ps;
    *pdf = fmaf(pdif, ORC_INV_PI * co, ((D * g1i) / (4.0f * ci)) * ps);
    float F, ct, eit, eti;
    fresnel_dielectric(vdot(wi, H), m->int_ior / m->ext_ior, &F, &ct, &eit, &eti);
    float G = g1i * mf_smith_g1(wo, H, au, av, beck);
    float spec = ((F * D) * G) / (4.0f * ci);
    float eta = m->int_ior / m->ext_ior, inv_eta_2 = 1.0f / (eta * eta);
    float dscale = (((ORC_INV_PI * inv_eta_2) * co) * t_i) * t_o;
    for (int k = 0; k < 3; ++k) {
        float diff = m->a[k] / (1.0f - ((m->flags & MTR_MAT_NONLINEAR) ? m->a[k] * m->internal_reflectance : m->internal_reflectance));
        val[k] = fmaf(diff, dscale, spec * m->c[k]);
    }
}
static int bsdf_is_smooth(const mtr_material *m) { return m->type == MTR_BSDF_DIFFUSE || bsdf_is_rough(m); }

/* eval_pdf: returns value (incl. cos) and pdf for a world-frame-local wo */
static void bsdf_eval_pdf(const mtr_material *m, v3 wi, v3 wo, float val[3], float *pdf)
{
    val[0] = val[1] = val[2] = 0.0f; *pdf = 0.0f;
    if (!bsdf_is_smooth(m)) return;
    if ((m->flags & MTR_MAT_TWOSIDED) && wi.z < 0.0f) { wi.z = -wi.z; wo.z = -wo.z; }
    if (bsdf_is_rough(m)) { rough_eval_pdf(m, wi, wo, val, pdf); return; }
    float ci = wi.z, co = wo.z;
    if (!(ci > 0.0f && co > 0.0f)) return;
    for (int k = 0; k < 3; ++k) val[k] = (m->a[k] * ORC_INV_PI) * co;
    *pdf = ORC_INV_PI * co;
}
static void bsdf_sample(const mtr_material *m, v3 wi, float u1, float ua, float ub, bsample *bs)
{
    memset(bs, 0, sizeof *bs); bs->eta = 1.0f;
    int flip = (m->flags & MTR_MAT_TWOSIDED) && wi.z < 0.0f;
    if (flip) wi.z = -wi.z;
    float ci = wi.z;
    switch (m->type) {
    case MTR_BSDF_DIFFUSE: {
        bs->wo = square_to_cos_hemi(ua, ub);
        bs->pdf = ORC_INV_PI * bs->wo.z;
        if (ci > 0.0f && bs->pdf > 0.0f) for (int k = 0; k < 3; ++k) bs->w[k] = m->a[k];
        break; }
    case MTR_BSDF_CONDUCTOR: {
        bs->wo = V(-wi.x, -wi.y, wi.z); bs->pdf = 1.0f; bs->delta = 1;
        if (ci > 0.0f) for (int k = 0; k < 3; ++k) bs->w[k] = m->c[k] * fresnel_conductor(ci, m->a[k], m->b[k]);
        break; }
    case MTR_BSDF_DIELECTRIC: {
        float r, ct, eit, eti;
        fresnel_dielectric(ci, m->int_ior / m->ext_ior, &r, &ct, &eit, &eti);
        int refl = u1 <= r;
        bs->delta = 1;
        bs->pdf = refl ? r : 1.0f - r;
        if (refl) { bs->wo = V(-wi.x, -wi.y, wi.z); for (int k = 0; k < 3; ++k) bs->w[k] = m->c[k]; }
        else { bs->wo = V(-eti * wi.x, -eti * wi.y, ct); bs->eta = eit;
               for (int k = 0; k < 3; ++k) bs->w[k] = m->c2[k] * (eti * eti); }
        break; }
    case MTR_BSDF_ROUGHCONDUCTOR: {          /* [RoughConductor::sample] */
        if (!(ci > 0.0f)) break;
        float pdf;
        v3 mm = ggx_sample(wi, m->alpha, rough_alpha_v(m), ua, ub, &pdf, (m->flags & MTR_MAT_BECKMANN) != 0u);
        float wim = vdot(wi, mm);
        v3 wo = V(fmaf(mm.x, 2.0f * wim, -wi.x), fmaf(mm.y, 2.0f * wim, -wi.y), fmaf(mm.z, 2.0f * wim, -wi.z));   /* reflect(wi, m) */
        bs->wo = wo;
        int ok = (pdf != 0.0f) && (wo.z > 0.0f);
        float weight = mf_smith_g1(wo, mm, m->alpha, rough_alpha_v(m), (m->flags & MTR_MAT_BECKMANN) != 0u);        /* sample_visible: weight = G1(wo) */
        bs->pdf = pdf / (4.0f * vdot(wo, mm));                /* Jacobian of the half-direction mapping */
        if (ok) for (int k = 0; k < 3; ++k) bs->w[k] = (fresnel_conductor(wim, m->a[k], m->b[k]) * weight) * m->c[k];
        break; }
    case MTR_BSDF_ROUGHPLASTIC: {            /* [RoughPlastic::sample]: lobe by sample1, then pdf() and eval() / pdf */
        if (!(ci > 0.0f)) break;
        float t_i = rough_transmittance(m, ci);
        float ps = (1.0f - t_i) * m->specular_sampling_weight, pdif = t_i * (1.0f - m->specular_sampling_weight);
        ps = ps / (ps + pdif);
        v3 wo;
        if (u1 < ps) {
            float pdf_m;
            v3 mm = ggx_sample(wi, m->alpha, m->alpha, ua, ub, &pdf_m, (m->flags & MTR_MAT_BECKMANN) != 0u);
            float wim = vdot(wi, mm);
            wo = V(fmaf(mm.x, 2.0f * wim, -wi.x), fmaf(mm.y, 2.0f * wim, -wi.y), fmaf(mm.z, 2.0f * wim, -wi.z));
        } else wo = square_to_cos_hemi(ua, ub);
        bs->wo = wo;
        float val[3], pdf;
        rough_eval_pdf(m, wi, wo, val, &pdf);
        bs->pdf = pdf;
        if (pdf > 0.0f) { float ip = 1.0f / pdf; for (int k = 0; k < 3; ++k) bs->w[k] = val[k] * ip; }
        break; }
    case MTR_BSDF_ROUGHDIELECTRIC: {         /* [RoughDielectric::sample]: visible normal for wi flipped up; reflect with probability F */
        if (ci == 0.0f) break;
        int beck = (m->flags & MTR_MAT_BECKMANN) != 0u;
        float au = m->alpha, av = rough_alpha_v(m);
        float pdf_m;
        v3 mm = ggx_sample(mulsign3(wi, ci), au, av, ua, ub, &pdf_m, beck);
        if (pdf_m == 0.0f) break;
        float wim = vdot(wi, mm);
        float F, ct, eit, eti;
        fresnel_dielectric(wim, m->int_ior / m->ext_ior, &F, &ct, &eit, &eti);
        int refl = u1 <= F;
        float pdf = pdf_m * (refl ? F : 1.0f - F);
        v3 wo; float w[3], dwh_dwo;
        if (refl) {
            wo = V(fmaf(mm.x, 2.0f * wim, -wi.x), fmaf(mm.y, 2.0f * wim, -wi.y), fmaf(mm.z, 2.0f * wim, -wi.z));
            bs->eta = 1.0f;
            for (int k = 0; k < 3; ++k) w[k] = m->c[k];
            dwh_dwo = 1.0f / (4.0f * vdot(wo, mm));
        } else {
            float kk = fmaf(wim, eti, ct);                    /* refract(wi, m, cos_theta_t, eta_ti) */
            wo = V(fmaf(mm.x, kk, -(wi.x * eti)), fmaf(mm.y, kk, -(wi.y * eti)), fmaf(mm.z, kk, -(wi.z * eti)));
            bs->eta = eit;
            float f2 = eti * eti;
            for (int k = 0; k < 3; ++k) w[k] = m->c2[k] * f2;
            float wom = vdot(wo, mm), t = fmaf(eit, wom, wim);
            dwh_dwo = ((eit * eit) * wom) / (t * t);
        }
        float g1 = mf_smith_g1(wo, mm, au, av, beck);
        bs->wo = wo;
        bs->pdf = pdf * fabsf(dwh_dwo);
        for (int k = 0; k < 3; ++k) bs->w[k] = w[k] * g1;
        break; }
    case MTR_BSDF_PLASTIC: {                 /* [Plastic::sample] */
        if (!(ci > 0.0f)) break;
        float eta = m->int_ior / m->ext_ior, inv_eta_2 = 1.0f / (eta * eta);
        float f_i, ct, eit, eti;
        fresnel_dielectric(ci, eta, &f_i, &ct, &eit, &eti);
        float ps, pdif;
        plastic_probs(m, f_i, &ps, &pdif);
        if (u1 < ps) {
            bs->wo = V(-wi.x, -wi.y, wi.z); bs->pdf = ps; bs->delta = 1;
            float s = f_i / ps;
            for (int k = 0; k < 3; ++k) bs->w[k] = m->c[k] * s;
        } else {
            v3 wo = square_to_cos_hemi(ua, ub);
            float f_o;
            fresnel_dielectric(wo.z, eta, &f_o, &ct, &eit, &eti);
            bs->wo = wo; bs->pdf = pdif * (ORC_INV_PI * wo.z);
            float scale = ((inv_eta_2 * (1.0f - f_i)) * (1.0f - f_o)) / pdif;
            if (bs->pdf > 0.0f)
                for (int k = 0; k < 3; ++k)
                    bs->w[k] = (m->a[k] / (1.0f - ((m->flags & MTR_MAT_NONLINEAR) ? m->a[k] * m->internal_reflectance : m->internal_reflectance))) * scale;
        }
        break; }
    case MTR_BSDF_THINDIELECTRIC: {          /* [ThinDielectric::sample]: r' = 2 r / (1 + r), straight-through transmission, eta = 1 */
        float r, ct, eit, eti;
        fresnel_dielectric(fabsf(wi.z), m->int_ior / m->ext_ior, &r, &ct, &eit, &eti);
        if (r < 1.0f) r *= 2.0f / (1.0f + r);
        int refl = u1 <= r;
        bs->delta = 1; bs->eta = 1.0f;
        bs->pdf = refl ? r : 1.0f - r;
        if (refl) { bs->wo = V(-wi.x, -wi.y, wi.z); for (int k = 0; k < 3; ++k) bs->w[k] = m->c[k]; }
        else { bs->wo = V(-wi.x, -wi.y, -wi.z); for (int k = 0; k < 3; ++k) bs->w[k] = m->c2[k]; }
        break; }
    default: break;
    }
    if (flip) bs->wo.z = -bs->wo.z;
}

/* test hooks: the BSDF functions on arrays of local directions */
void orc_bsdf_eval_pdf(const mtr_material *m, uint32_t n, const float *wi3, const float *wo3, float *val3, float *pdf)
{
    for (uint32_t i = 0; i < n; ++i)
        bsdf_eval_pdf(m, V(wi3[3 * i], wi3[3 * i + 1], wi3[3 * i + 2]), V(wo3[3 * i], wo3[3 * i + 1], wo3[3 * i + 2]), val3 + 3 * i, pdf + i);
}
void orc_bsdf_sample(const mtr_material *m, uint32_t n, const float *wi3, const float *u1, const float *ua, const float *ub,
                     float *wo3, float *pdf, float *w3)
{
    for (uint32_t i = 0; i < n; ++i) {
        bsample bs;
        bsdf_sample(m, V(wi3[3 * i], wi3[3 * i + 1], wi3[3 * i + 2]), u1[i], ua[i], ub[i], &bs);
        wo3[3 * i] = bs.wo.x; wo3[3 * i + 1] = bs.wo.y; wo3[3 * i + 2] = bs.wo.z; pdf[i] = bs.pdf;
        for (int k = 0; k < 3; ++k) w3[3 * i + k] = bs.w[k];
    }
}

/* ------------------------------------------------------------------ */
/* Film  (transient_hdr_film.py:250-276, transient_image_block.py:103-151) */
/* ------------------------------------------------------------------ */
typedef struct { uint64_t closest, shadow, bounces, splats; } lane_counters;
typedef struct {
    const mtr_film_desc *f; float *transient; float *steady;
    orc_splat_rec *log; uint64_t log_cap; uint64_t *log_n;
} film_t;

int orc_bin_index(float distance, float start_opl, float bin_width_opl, uint32_t T)
{
    float pos_distance = (distance - start_opl) / bin_width_opl;        /* transient_hdr_film.py:263 */
    if (!(pos_distance >= 0.0f && pos_distance < (float)T)) return -1;  /* :265 */
    return (int)(uint32_t)floorf(pos_distance);                          /* transient_image_block.py:132 */
}
/* flat element index of a contribution (transient_image_block.py:134-144), without the channel */
static size_t film_cell(const mtr_film_desc *f, uint32_t x, uint32_t y, uint32_t laser_x, uint32_t laser_y, uint32_t bin)
{
    size_t index = (size_t)y * f->width + x;
    if (f->laser_scan_width && f->laser_scan_height) {                   /* exhaustive_scan :134-140 */
        index = index * f->laser_scan_width + laser_x;
        index = index * f->laser_scan_height + laser_y;
    }
    return index * f->temporal_bins + bin;                               /* :139 / :143 */
}
static size_t film_cells(const mtr_film_desc *f)
{
    size_t n = (size_t)f->width * f->height * f->temporal_bins;
    if (f->laser_scan_width && f->laser_scan_height) n *= (size_t)f->laser_scan_width * f->laser_scan_height;
    return n;
}
static void add_transient_l(film_t *F, uint32_t px, uint32_t py, float distance, const float spec[3],
                            float sample_scale, uint32_t lane, uint32_t depth, uint32_t kind, lane_counters *C,
                            uint32_t laser_x, uint32_t laser_y);
static void add_transient(film_t *F, uint32_t px, uint32_t py, float distance, const float spec[3],
                          float sample_scale, uint32_t lane, uint32_t depth, uint32_t kind, lane_counters *C)
{
    add_transient_l(F, px, py, distance, spec, sample_scale, lane, depth, kind, C, 0, 0);
}
static void add_transient_l(film_t *F, uint32_t px, uint32_t py, float distance, const float spec[3],
                            float sample_scale, uint32_t lane, uint32_t depth, uint32_t kind, lane_counters *C,
                            uint32_t laser_x, uint32_t laser_y)
{
    const mtr_film_desc *f = F->f;
    /* common.py:417-421: spec * sample_scale, then * ray_weight (== 1) */
    float val[3] = { spec[0] * sample_scale, spec[1] * sample_scale, spec[2] * sample_scale };
    if (f->n_frequencies) {                        /* phasor_hdr_film.add_transient_data (:240-262) + PhasorImageBlock.put (:42-67) */
        if (!isfinite(distance)) return;                                    /* active &= isfinite(opl) :47 */
        uint32_t x = px - f->crop_offset_x, y = py - f->crop_offset_y;
        if (!(x < f->width && y < f->height)) return;
        if (val[0] == 0.0f) return;                                         /* spec.x (monochromatic); adding +0 is a no-op */
        const float rel = distance - f->start_opl;                          /* :249 */
        float *dst = F->transient + ((size_t)y * f->width + x) * (2u * (size_t)f->n_frequencies + 1u);
        for (uint32_t k = 0; k < f->n_frequencies; ++k) {
            float c, sn;
            phasor_term(f->frequencies[k], rel, &c, &sn);
            const float re = val[0] * c, im = val[0] * sn;
#pragma omp atomic
            dst[2 * k] += re;
#pragma omp atomic
            dst[2 * k + 1] += im;
        }
        C->splats += 1;
        return;
    }
    int bin = orc_bin_index(distance, f->start_opl, f->bin_width_opl, f->temporal_bins);
    if (bin < 0) return;
    /* p = floor(pos) - offset (transient_image_block.py:132); pos carries the crop offset */
    uint32_t x = px - f->crop_offset_x, y = py - f->crop_offset_y;
    if (!(x < f->width && y < f->height)) return;                        /* :146 */
    if (val[0] == 0.0f && val[1] == 0.0f && val[2] == 0.0f) return;     /* adding +0 is a no-op */
    if (f->laser_scan_width && !(laser_x < f->laser_scan_width && laser_y < f->laser_scan_height)) return;
    size_t index = film_cell(f, x, y, laser_x, laser_y, (uint32_t)bin) * 4u;             /* :134-144 */
    for (int k = 0; k < 3; ++k) {
#pragma omp atomic
        F->transient[index + k] += val[k];                               /* accum: scatter_reduce(Add) :79-81 */
    }
    /* channel 3 ("W") receives alpha = 0.0 (transient_hdr_film.py:269-272): stays 0 */
    C->splats += 1;
    if (F->log) {
        uint64_t i;
#pragma omp atomic capture
        i = (*F->log_n)++;
        if (i < F->log_cap) {
            orc_splat_rec *R = &F->log[i];
            R->lane = lane; R->depth_kind = depth | (kind << 16); R->pixel = y * f->width + x;
            R->bin = (uint32_t)(film_cell(f, 0, 0, laser_x, laser_y, (uint32_t)bin));     /* position in the pixel's row */
            R->r = val[0]; R->g = val[1]; R->b = val[2]; R->opl = distance;
        }
    }
}

/* ------------------------------------------------------------------ */
/* Sensor  [mitsuba3: ADIntegrator.sample_rays (python/ad/integrators/common.py),
 *          src/sensors/perspective.cpp sample_ray]                    */
/* ------------------------------------------------------------------ */
static ray3 sample_ray(const mtr_scene_desc *d, uint32_t px, uint32_t py, float j1, float j2)
{
    const mtr_film_desc *f = &d->film; const mtr_camera *c = &d->camera;
    float pos_x = (float)px + j1, pos_y = (float)py + j2;                /* pos_f = pos + next_2d() */
    float scale_x = 1.0f / (float)f->crop_width, scale_y = 1.0f / (float)f->crop_height;
    float off_x = -(float)f->crop_offset_x * scale_x, off_y = -(float)f->crop_offset_y * scale_y;
    float sx = fmaf(pos_x, scale_x, off_x), sy = fmaf(pos_y, scale_y, off_y);
    const float *M = c->sample_to_camera;
    float nx = fmaf(M[0], sx, fmaf(M[1], sy, M[3]));
    float ny = fmaf(M[4], sx, fmaf(M[5], sy, M[7]));
    float nz = fmaf(M[8], sx, fmaf(M[9], sy, M[11]));
    float nw = fmaf(M[12], sx, fmaf(M[13], sy, M[15]));
    float iw = 1.0f / nw;
    v3 dl = vnormalize(V(nx * iw, ny * iw, nz * iw));
    const float *T = c->to_world;
    ray3 r;
    r.d = V(fmaf(T[0], dl.x, fmaf(T[1], dl.y, T[2] * dl.z)),
            fmaf(T[4], dl.x, fmaf(T[5], dl.y, T[6] * dl.z)),
            fmaf(T[8], dl.x, fmaf(T[9], dl.y, T[10] * dl.z)));
    float inv_z = 1.0f / dl.z;
    float near_t = c->near_clip * inv_z, far_t = c->far_clip * inv_z;
    r.o = vfma(r.d, near_t, V(T[3], T[7], T[11]));
    r.maxt = far_t - near_t;
    return r;
}

/* [mitsuba3: mis_weight in python/ad/integrators/common.py] */
static float mis_weight(float a, float b)
{
    float a2 = a * a, b2 = b * b;
    float w = a2 / (a2 + b2);
    return isfinite(w) ? w : 0.0f;
}

/* ------------------------------------------------------------------ */
/* One lane of TransientPath.sample (transientpath.py:88-326)          */
/* ------------------------------------------------------------------ */


static void trace_lane(const orc_scene *sc, const mtr_render_params *P, film_t *F, uint32_t lane,
                       int use_bvh, lane_counters *C)
{
    const mtr_scene_desc *d = sc->d; const mtr_film_desc *f = &d->film;
    const uint32_t spp = P->spp_total;
    const float sample_scale = (float)(1.0 / (double)(P->spp_scale ? P->spp_scale : spp));               /* common.py:173-175: python float 1.0/total_spp */
    const uint32_t max_depth = P->max_depth < 0 ? 0xffffffffu : (uint32_t)P->max_depth;
    const uint32_t rr_depth = (uint32_t)P->rr_depth;

    /* sample_rays: lane -> pixel */
    uint32_t idx = lane / spp;
    uint32_t py = idx / f->crop_width, px = idx - f->crop_width * py;
    px += f->crop_offset_x; py += f->crop_offset_y;

    pcg32 rng; sampler_seed_v(&rng, P->seed, lane, P->flags);   /* common.py:52 */
    float j1 = pcg32_next_f32(&rng), j2 = pcg32_next_f32(&rng);
    ray3 ray = sample_ray(d, px, py, j1, j2);

    /* transientpath.py:118-131 */
    uint32_t depth = 0; float L[3] = { 0, 0, 0 }, beta[3] = { 1, 1, 1 };   /* β_init == 1, utils.py:9-21 */
    float eta = 1.0f, distance = 0.0f;
    int active = 1, prev_delta = 1; v3 prev_p = V(0, 0, 0); float prev_pdf = 1.0f;

    if (P->flags & MTR_FLAG_CAMERA_UNWARP) {                             /* :133-138 */
        hit_t h = intersect(sc, &ray, use_bvh); C->closest++;
        if (h.prim >= 0) distance = -h.t;
    }

    while (active) {                                                     /* :140 */
        C->bounces++;
        int active_next = 1;
        hit_t h = intersect(sc, &ray, use_bvh); C->closest++;            /* :148-151 */
        sinter si = make_si(sc, &ray, h);
        distance += si.t * eta;                                          /* :154 (inf on a miss) */
        mtr_material mat_copy;
        const mtr_material *mat = si.valid ? material_at(d, &d->materials[d->tri_material[si.prim]], &si, &mat_copy) : NULL;
        int em = si.valid ? d->tri_emitter[si.prim] : -1;

        /* ---- direct emission :166-176 ---- */
        float Le[3] = { 0, 0, 0 };
        if (em >= 0 && !(P->flags & MTR_FLAG_DISCARD_DIRECT_LIGHT)) {
            /* ds = DirectionSample3f(scene, si, ref=prev_si) */
            v3 rel = vsub(si.p, prev_p);
            float dist = sqrtf(vdot(rel, rel));
            v3 dd = vdivs(rel, dist);
            /* pdf_emitter_direction(prev_si, ds, ~prev_bsdf_delta) [AreaLight::pdf_direction, Shape::pdf_direction] */
            float em_pdf = 0.0f;
            if (!prev_delta) {
                float dp = vdot(dd, si.n);          /* DirectionSample3f(scene, si, ref): ds.n = si.sh_frame.n [PositionSample(si)] */
                if (dp < 0.0f) {
                    float adp = fabsf(dp);
                    em_pdf = sc->em_inv_area[em] * (adp != 0.0f ? (dist * dist) / adp : 0.0f);
                    if (d->n_emitters > 1) em_pdf *= 1.0f / (float)d->n_emitters;
                }
            }
            float mis = mis_weight(prev_pdf, em_pdf);
            /* emitter.eval(si): radiance where cos_theta(si.wi) > 0 [AreaLight::eval] */
            if (si.wi.z > 0.0f)
                for (int k = 0; k < 3; ++k) Le[k] = (beta[k] * mis) * d->emitters[em].radiance[k];
        }
        add_transient(F, px, py, distance, Le, sample_scale, lane, depth, 0, C);   /* :179-180 */

        /* ---- emitter sampling :185-218 ---- */
        active_next &= (depth + 1 < max_depth) && si.valid;
        int active_em = active_next && bsdf_is_smooth(mat);
        float u1 = pcg32_next_f32(&rng), u2 = pcg32_next_f32(&rng);     /* sampler.next_2d() :193 */
        float Lr[3] = { 0, 0, 0 }; float ds_dist = 0.0f;
        if (active_em && d->n_emitters > 0) {
            /* [Scene::sample_emitter_direction] */
            uint32_t ei = 0; float pmf = 1.0f;
            if (d->n_emitters > 1) {
                float ne = (float)d->n_emitters;
                float su = u1 * ne;
                uint32_t i = (uint32_t)su; if (i > d->n_emitters - 1) i = d->n_emitters - 1;
                ei = i; u1 = su - (float)i; pmf = 1.0f / ne;
            }
            const mtr_emitter *E = &d->emitters[ei];
            v3 ep, en;
            if (E->is_mesh) {                   /* [Mesh::sample_position]: face by area (reusing sample.y), uniform triangle */
                float sy = u2, r2, fp; uint32_t fi = 0;
                if (E->n_tris > 1) { fi = distr_sample_reuse(sc->em_face_cdf + E->first_tri, sc->em_face_pmf + E->first_tri, E->n_tris, u2, &r2, &fp); sy = r2; }
                const orc_tri *T = &sc->tris[E->first_tri + fi];
                float tt = sqrtf(fmaxf(1.0f - u1, 0.0f));
                float b0 = 1.0f - tt, b1 = tt * sy;
                ep = V(fmaf(T->e1.x, b0, fmaf(T->e2.x, b1, T->p0.x)), fmaf(T->e1.y, b0, fmaf(T->e2.y, b1, T->p0.y)),
                       fmaf(T->e1.z, b0, fmaf(T->e2.z, b1, T->p0.z)));
                en = T->n;
                if (T->smooth) {                /* has_vertex_normals: ps.n = normalize(fmadd(n0, 1 - b.x - b.y, fmadd(n1, b.x, n2 * b.y))) */
                    const float *vn = d->tri_normals + 9 * (size_t)(E->first_tri + fi);
                    float w0 = 1.0f - b0 - b1;
                    en = vnormalize(V(fmaf(vn[0], w0, fmaf(vn[3], b0, vn[6] * b1)), fmaf(vn[1], w0, fmaf(vn[4], b0, vn[7] * b1)),
                                      fmaf(vn[2], w0, fmaf(vn[5], b0, vn[8] * b1))));
                }
            } else {                            /* [Rectangle::sample_position] */
                float a = fmaf(u1, 2.0f, -1.0f), b = fmaf(u2, 2.0f, -1.0f);
                ep = V(fmaf(E->du[0], a, fmaf(E->dv[0], b, E->center[0])),
                       fmaf(E->du[1], a, fmaf(E->dv[1], b, E->center[1])),
                       fmaf(E->du[2], a, fmaf(E->dv[2], b, E->center[2])));
                en = sc->em_n[ei];
            }
            /* [Shape::sample_direction] */
            v3 dd = vsub(ep, si.p);
            float dist2 = vdot(dd, dd), dist = sqrtf(dist2);
            dd = vdivs(dd, dist);
            float dp = vdot(dd, en), adp = fabsf(dp);
            float x = dist2 / adp;
            float pdf_dir = sc->em_inv_area[ei] * (isfinite(x) ? x : 0.0f);
            ds_dist = dist;
            /* [AreaLight::sample_direction] active &= dot(d,n) < 0 && pdf != 0; spec = radiance / pdf */
            int ok = (dp < 0.0f) && (pdf_dir != 0.0f);
            float emw[3] = { 0, 0, 0 };
            if (ok) { float ip = 1.0f / pdf_dir; for (int k = 0; k < 3; ++k) emw[k] = E->radiance[k] * ip; }
            float pdf = pdf_dir;
            if (d->n_emitters > 1) {                                    /* ds.pdf *= pmf; spec *= 1/pmf */
                pdf = pdf_dir * pmf;
                for (int k = 0; k < 3; ++k) emw[k] *= (float)d->n_emitters;
            }
            int active_e = active_em && (pdf != 0.0f) && ok;            /* :194 (zero-weight lanes skipped) */
            /* visibility [Interaction::spawn_ray_to + Scene::ray_test] */
            if (active_e) {
                v3 o = offset_p(&si, vsub(ep, si.p));
                v3 sd = vsub(ep, o);
                float sdist = sqrtf(vdot(sd, sd));
                ray3 sr; sr.o = o; sr.d = vdivs(sd, sdist);
                sr.maxt = sdist * (1.0f - ORC_SHADOW_EPS);
                C->shadow++;
                if (ray_test(sc, &sr, use_bvh)) { emw[0] = emw[1] = emw[2] = 0.0f; }
                /* :207-213 */
                v3 wo = to_local(&si, dd);
                float bv[3], bpdf; bsdf_eval_pdf(mat, si.wi, wo, bv, &bpdf);
                float mis_em = mis_weight(pdf, bpdf);                   /* ds.delta == false for area lights */
                for (int k = 0; k < 3; ++k) Lr[k] = ((beta[k] * mis_em) * bv[k]) * emw[k];
            }
        }
        add_transient(F, px, py, distance + ds_dist * eta, Lr, sample_scale, lane, depth, 1, C);   /* :216-218 */

        /* ---- BSDF sampling :222-233 ---- */
        float s1 = pcg32_next_f32(&rng);
        float s2a = pcg32_next_f32(&rng), s2b = pcg32_next_f32(&rng);
        bsample bs; memset(&bs, 0, sizeof bs); bs.eta = 1.0f;
        if (active_next) bsdf_sample(mat, si.wi, s1, s2a, s2b, &bs);
        for (int k = 0; k < 3; ++k) L[k] = (L[k] + Le[k]) + Lr[k];      /* :230 */
        if (active_next) {
            v3 wo_w = to_world(&si, bs.wo);
            ray.o = offset_p(&si, wo_w); ray.d = wo_w; ray.maxt = INFINITY;   /* si.spawn_ray :231 */
        }
        eta *= bs.eta;                                                   /* :232 */
        for (int k = 0; k < 3; ++k) beta[k] *= bs.w[k];                  /* :233 */
        prev_p = si.p; prev_pdf = bs.pdf; prev_delta = bs.delta;   /* :237-240 */

        /* ---- stopping criterion :245-257 ---- */
        float bmax = fmaxf(beta[0], fmaxf(beta[1], beta[2]));
        active_next &= (bmax != 0.0f);
        float rr_prob = fminf(bmax * (eta * eta), 0.95f);
        active_next &= rr_prob > 0.0f;
        int rr_active = depth >= rr_depth;
        if (rr_active) {
            float inv = rr_prob > 0.0f ? 1.0f / rr_prob : 0.0f;
            for (int k = 0; k < 3; ++k) beta[k] *= inv;
        }
        float rr_u = pcg32_next_f32(&rng);                               /* :256 */
        int rr_continue = rr_u < rr_prob;
        active_next &= (!rr_active) || rr_continue;

        if (si.valid) depth += 1;                                        /* :318 */
        active = active_next;                                            /* :319 */
    }
    /* steady splat: block.put(pos, [L.r, L.g, L.b, 1]) common.py:187-200 */
    if (F->steady) {
        uint32_t x = px - f->crop_offset_x, y = py - f->crop_offset_y;
        if (x < f->width && y < f->height) {
            size_t i = ((size_t)y * f->width + x) * 4u;
            for (int k = 0; k < 3; ++k) {
#pragma omp atomic
                F->steady[i + k] += L[k];
            }
#pragma omp atomic
            F->steady[i + 3] += 1.0f;
        }
    }
}

/* ================================================================== */
/* NLOS tier: TransientNLOSPath (mitransient/integrators/transientnlospath.py)
 * + NLOSCaptureMeter (mitransient/sensors/nloscapturemeter.py) + mitsuba's `projector`.
 * Same status as above: the reference's Python is restated literally, the Mitsuba pieces
 * (projector.sample_direction, Rectangle/Mesh.sample_position, DiscreteDistribution) from
 * their published algorithms [upstream-unverified].  PARITY UNPINNED.            */
/* ================================================================== */
#define ORC_EPS 5.9604644775390625e-8f      /* dr.epsilon(Float) = 2^-24 */

typedef struct {
    const mtr_nlos_desc *n;
    /* projector */
    v3 l_origin, l_forward; float l_inv[12];      /* rows of the inverse rigid transform (world -> local) */
    float l_cot;                                   /* 1 / tan(fov/2) */
    /* relay wall rectangle */
    v3 w_center, w_du, w_dv;
    /* hidden-geometry distribution over shapes, per-shape face distributions (normalised, f32) */
    float *shape_pmf, *shape_cdf;
    float *face_cdf, *face_pmf;                    /* per triangle, within its shape */
    float *shape_inv_area;
    v3 *rect_n;                                    /* rectangle shapes: normal */
    /* scan */
    v3 *sensor_targets; v3 laser_target_single;
    v3 *laser_targets; uint32_t laser_w, laser_h;  /* Exhaustive: illuminated points (:340-381) */
} nlos_scene;

#define tri_area_d orc_tri_area_d

/* [mitsuba3: DiscreteDistribution::sample_reuse_pmf] on a normalised f32 cdf/pmf table */
static uint32_t distr_sample_reuse(const float *cdf, const float *pmf, uint32_t n, float value, float *reused, float *pmf_out)
{
    uint32_t i = 0;
    while (i + 1 < n && !(value < cdf[i])) ++i;
    while (i + 1 < n && pmf[i] == 0.0f) ++i;
    float prev = i ? cdf[i - 1] : 0.0f;
    *reused = (value - prev) / pmf[i];
    *pmf_out = pmf[i];
    return i;
}

/* [mitsuba3: Projector::sample_direction]; returns the weight (spec), ds.dist, ds.d */
static void projector_sample(const nlos_scene *N, v3 p, float w[3], float *dist, v3 *dir_to_emitter)
{
    const float *M = N->l_inv;
    v3 rel = vsub(p, N->l_origin);
    v3 loc = V(vdot(V(M[0], M[1], M[2]), rel), vdot(V(M[4], M[5], M[6]), rel), vdot(V(M[8], M[9], M[10]), rel));
    float iz = 1.0f / loc.z;
    float uvx = 0.5f - (0.5f * N->l_cot) * (loc.x * iz), uvy = 0.5f - (0.5f * N->l_cot) * (loc.y * iz);
    int ok = (uvx >= 0.0f && uvx <= 1.0f && uvy >= 0.0f && uvy <= 1.0f && loc.z > 0.0f);
    v3 d = vsub(N->l_origin, p);
    float d2 = vdot(d, d);
    *dist = sqrtf(d2);
    d = vdivs(d, *dist);
    *dir_to_emitter = d;
    /* spec *= pi * scale / (z^2 * -dot(n, d)) : irradiance at z = 1 on the axis */
    float f = (ORC_PI * N->n->laser_scale) * (iz * iz) / -vdot(N->l_forward, d);
    for (int k = 0; k < 3; ++k) w[k] = ok ? N->n->laser_irradiance[k] * f : 0.0f;
}

/* rectangle.sample_position [mitsuba3: Rectangle::sample_position] */
static v3 rect_point(v3 c, v3 du, v3 dv, float u, float v)
{
    float a = fmaf(u, 2.0f, -1.0f), b = fmaf(v, 2.0f, -1.0f);
    return V(fmaf(du.x, a, fmaf(dv.x, b, c.x)), fmaf(du.y, a, fmaf(dv.y, b, c.y)), fmaf(du.z, a, fmaf(dv.z, b, c.z)));
}

/* NLOSCaptureMeter.sample_ray (nloscapturemeter.py:136-180) for film sample (sx, sy) in [0,1)^2 */
static ray3 nlos_sensor_ray(const nlos_scene *N, const mtr_film_desc *f, float sx, float sy)
{
    float W = (float)f->width, H = (float)f->height;
    float gx = (floorf(sx * W) + 0.5f) / W, gy = (floorf(sy * H) + 0.5f) / H;     /* :146-149 pixel centre */
    v3 target = rect_point(N->w_center, N->w_du, N->w_dv, gx, gy);
    if (N->n->sensor_is_confocal)                                                  /* :142 target = self.laser_target */
        target = V(N->n->sensor_target[0], N->n->sensor_target[1], N->n->sensor_target[2]);
    v3 o = V(N->n->sensor_origin[0], N->n->sensor_origin[1], N->n->sensor_origin[2]);
    v3 dir = vsub(target, o);
    float dist = sqrtf(vdot(dir, dir));
    ray3 r; r.o = o; r.d = vdivs(dir, dist); r.maxt = INFINITY;
    return r;
}

static void nlos_build(nlos_scene *N, const orc_scene *sc, int use_bvh)
{
    const mtr_scene_desc *d = sc->d; const mtr_nlos_desc *n = d->nlos;
    memset(N, 0, sizeof *N); N->n = n;
    const float *T = n->laser_to_world;
    N->l_origin = V(T[3], T[7], T[11]);
    N->l_forward = V(T[2], T[6], T[10]);
    /* rigid: inverse rotation = transpose */
    float inv[12] = { T[0], T[4], T[8], 0, T[1], T[5], T[9], 0, T[2], T[6], T[10], 0 };
    memcpy(N->l_inv, inv, sizeof inv);
    N->l_cot = (float)(1.0 / tan(0.5 * (double)n->laser_fov * 3.14159265358979323846 / 180.0));
    static const mtr_shape no_wall = { 0, 0, 1, { 0, 0, 0 }, { 1, 0, 0 }, { 0, 1, 0 } };
    const mtr_shape *rw = n->relay_shape == MTR_NLOS_NO_RELAY ? &no_wall : &n->shapes[n->relay_shape];   /* perspective camera: no relay wall */
    N->w_center = V(rw->center[0], rw->center[1], rw->center[2]);
    N->w_du = V(rw->du[0], rw->du[1], rw->du[2]); N->w_dv = V(rw->dv[0], rw->dv[1], rw->dv[2]);
    uint32_t ns = n->n_shapes;
    N->shape_pmf = calloc(ns, 4); N->shape_cdf = calloc(ns, 4); N->shape_inv_area = calloc(ns, 4);
    N->rect_n = calloc(ns, sizeof(v3));
    N->face_cdf = calloc(d->n_tris ? d->n_tris : 1, 4); N->face_pmf = calloc(d->n_tris ? d->n_tris : 1, 4);
    double *area = calloc(ns, sizeof(double)), total = 0.0;
    for (uint32_t s = 0; s < ns; ++s) {
        const mtr_shape *S = &n->shapes[s];
        double a = 0.0;
        for (uint32_t t = 0; t < S->n_tris; ++t) a += tri_area_d(d->tri_verts + 9 * (size_t)(S->first_tri + t));
        double acc = 0.0;
        for (uint32_t t = 0; t < S->n_tris; ++t) {
            double at = tri_area_d(d->tri_verts + 9 * (size_t)(S->first_tri + t));
            acc += at;
            N->face_pmf[S->first_tri + t] = (float)(at / a);
            N->face_cdf[S->first_tri + t] = (float)(acc / a);
        }
        if (S->is_rectangle) {
            v3 du = V(S->du[0], S->du[1], S->du[2]), dv = V(S->dv[0], S->dv[1], S->dv[2]);
            v3 c = vcross(du, dv);
            double len = sqrt((double)c.x * c.x + (double)c.y * c.y + (double)c.z * c.z);
            a = 4.0 * len;
            N->rect_n[s] = vdivs(c, sqrtf(vdot(c, c)));
            if (S->is_rectangle & MTR_RECT_FLIP_NORMALS) N->rect_n[s] = V(-N->rect_n[s].x, -N->rect_n[s].y, -N->rect_n[s].z);
        }
        N->shape_inv_area[s] = (float)(1.0 / a);
        /* transientnlospath.py:277-292: the relay wall has weight 0 unless ..._includes_relay_wall */
        area[s] = (s == n->relay_shape && !(n->flags & MTR_NLOS_HG_INCLUDES_WALL)) ? 0.0 : a;
        total += area[s];
    }
    double acc = 0.0;
    for (uint32_t s = 0; s < ns; ++s) {
        acc += area[s];
        N->shape_pmf[s] = total > 0.0 ? (float)(area[s] / total) : 0.0f;
        N->shape_cdf[s] = total > 0.0 ? (float)(acc / total) : 0.0f;
    }
    free(area);
    /* scanned points: one ray per film pixel (transientnlospath.py:295-312) */
    const mtr_film_desc *f = &d->film;
    N->sensor_targets = calloc((size_t)f->width * f->height, sizeof(v3));
    for (uint32_t y = 0; y < f->height; ++y)
        for (uint32_t x = 0; x < f->width; ++x) {
            /* sensor.sample_ray at the film sample (x / W, y / H) (:296-308): nlos_capture_meter, or the scene's camera */
            ray3 r = n->relay_shape == MTR_NLOS_NO_RELAY ? sample_ray(d, x, y, 0.0f, 0.0f)
                                                         : nlos_sensor_ray(N, f, (float)x / (float)f->width, (float)y / (float)f->height);
            hit_t h = intersect(sc, &r, use_bvh);
            sinter si = make_si(sc, &r, h);
            N->sensor_targets[(size_t)y * f->width + x] = si.p;          /* (0,0,0) on a miss, like zeros(si) */
        }
    /* Single capture: where the laser's optical axis meets the scene (:328-336) */
    ray3 lr; lr.o = N->l_origin; lr.d = N->l_forward; lr.maxt = INFINITY;
    hit_t lh = intersect(sc, &lr, use_bvh);
    sinter lsi = make_si(sc, &lr, lh);
    N->laser_target_single = lsi.p;
    if (n->capture_type == MTR_CAPTURE_EXHAUSTIVE) {
        N->laser_w = f->laser_scan_width; N->laser_h = f->laser_scan_height;
        const uint32_t nl = N->laser_w * N->laser_h;
        N->laser_targets = calloc(nl ? nl : 1, sizeof(v3));
        if (n->flags & MTR_NLOS_FORCE_EQUAL_GRIDS) {                     /* :341-346 (grids checked equal by the caller) */
            for (uint32_t i = 0; i < nl && i < f->width * f->height; ++i) N->laser_targets[i] = N->sensor_targets[i];
        } else {
            /* dummy projector with illumination_scan_fov (:347-381); meshgrid 'xy': point i = (x = i % Lx, y = i / Lx).
             * [mitsuba3: Projector::sample_ray, constant irradiance]: uv = sample3,
             * near_p = sample_to_camera * (u, v, 0) = ((1-2u) tan(fov/2), (1-2v) tan(fov/2), 1) * near, d = to_world * normalize(near_p) */
            const float th = (float)tan(0.5 * (double)n->illumination_scan_fov * 3.14159265358979323846 / 180.0);
            const float *T = n->laser_to_world;
            for (uint32_t i = 0; i < nl; ++i) {
                uint32_t y = i / N->laser_w, x = i - y * N->laser_w;
                float u = (float)x / (float)N->laser_w, v = (float)y / (float)N->laser_h;
                v3 loc = vnormalize(V((1.0f - 2.0f * u) * th, (1.0f - 2.0f * v) * th, 1.0f));
                ray3 r; r.o = N->l_origin; r.maxt = INFINITY;
                r.d = V(fmaf(T[0], loc.x, fmaf(T[1], loc.y, T[2] * loc.z)), fmaf(T[4], loc.x, fmaf(T[5], loc.y, T[6] * loc.z)),
                        fmaf(T[8], loc.x, fmaf(T[9], loc.y, T[10] * loc.z)));
                hit_t h = intersect(sc, &r, use_bvh);
                sinter si = make_si(sc, &r, h);
                N->laser_targets[i] = si.p;
            }
        }
    }
}
static void nlos_free(nlos_scene *N)
{
    free(N->shape_pmf); free(N->shape_cdf); free(N->shape_inv_area); free(N->rect_n);
    free(N->face_cdf); free(N->face_pmf); free(N->sensor_targets); free(N->laser_targets);
}

/* si.spawn_ray_to(t) [mitsuba3: Interaction::spawn_ray_to] */
static ray3 spawn_ray_to(const sinter *si, v3 t)
{
    v3 o = offset_p(si, vsub(t, si->p));
    v3 dd = vsub(t, o);
    float dist = sqrtf(vdot(dd, dd));
    ray3 r; r.o = o; r.d = vdivs(dd, dist); r.maxt = dist * (1.0f - ORC_SHADOW_EPS);
    return r;
}

/* bsdf.eval(ctx, si, wo): value * cos (diffuse only is smooth) */
static void bsdf_eval(const mtr_material *m, v3 wi, v3 wo, float val[3])
{
    float pdf; bsdf_eval_pdf(m, wi, wo, val, &pdf);
}

/* emitter_nee_sample (transientnlospath.py:432-509).  `distance` by value: the caller's copy is not changed */
static void nlos_emitter_nee(const orc_scene *sc, const nlos_scene *N, film_t *F, pcg32 *rng, const sinter *si,
                             const mtr_material *mat, const float beta[3], float distance, float eta, uint32_t depth,
                             int active_e, int focus_laser, uint32_t px, uint32_t py, float sample_scale, uint32_t lane,
                             uint32_t loop_depth, int use_bvh, lane_counters *C, float Lr[3], uint32_t laser_x, uint32_t laser_y)
{
    const mtr_nlos_desc *n = N->n;
    Lr[0] = Lr[1] = Lr[2] = 0.0f;
    if (active_e) {                                                  /* :441 visibility of the emitter origin */
        ray3 sr = spawn_ray_to(si, N->l_origin);
        C->shadow++;
        if (ray_test(sc, &sr, use_bvh)) active_e = 0;
    }
    if (!active_e) return;                                           /* masked lanes draw nothing: next_2d(active_e) */
    float u1 = pcg32_next_f32(rng), u2 = pcg32_next_f32(rng); (void)u1; (void)u2;
    float w[3], ds_dist; v3 dir;
    if (focus_laser && (n->capture_type == MTR_CAPTURE_CONFOCAL || n->capture_type == MTR_CAPTURE_EXHAUSTIVE)) {   /* :448-458 */
        v3 rel = vsub(N->l_origin, si->p);
        float dist_e = sqrtf(vdot(rel, rel));
        v3 pf = vfma(N->l_forward, dist_e, N->l_origin);
        projector_sample(N, pf, w, &ds_dist, &dir);
    } else {
        projector_sample(N, si->p, w, &ds_dist, &dir);
    }
    /* wo = to_local(normalize(ds.p - si.p)) :483 */
    v3 wo = to_local(si, vnormalize(vsub(N->l_origin, si->p)));
    float bv[3], bpdf; bsdf_eval_pdf(mat, si->wi, wo, bv, &bpdf);
    if (n->filter_depth != -1) active_e = active_e && (depth == (uint32_t)n->filter_depth);       /* :489-490 */
    if (n->flags & MTR_NLOS_DISCARD_DIRECT) active_e = active_e && (depth > 2);                   /* :491-492 */
    if (!active_e) return;
    for (int k = 0; k < 3; ++k) Lr[k] = (beta[k] * bv[k]) * w[k];                                /* :493 */
    if (n->flags & MTR_NLOS_ACCOUNT_FIRST_LAST) distance += ds_dist * eta;                        /* :497-498 */
    if (n->capture_type != MTR_CAPTURE_EXHAUSTIVE) laser_x = laser_y = 0;                         /* :502-505 */
    add_transient_l(F, px, py, distance, Lr, sample_scale, lane, loop_depth, 1, C, laser_x, laser_y);   /* :506-507 */
}

/* emitter_laser_targets_sample (transientnlospath.py:511-564) */
static void nlos_laser_targets(const orc_scene *sc, const nlos_scene *N, film_t *F, pcg32 *rng, const sinter *si,
                               const mtr_material *mat, v3 lt, const float beta[3], float distance, float eta,
                               uint32_t depth, int active_e, uint32_t px, uint32_t py, float sample_scale,
                               uint32_t lane, uint32_t loop_depth, int use_bvh, lane_counters *C, float Lr[3],
                               uint32_t laser_x, uint32_t laser_y)
{
    Lr[0] = Lr[1] = Lr[2] = 0.0f;
    if (!active_e) return;
    v3 dd = vsub(lt, si->p);
    float dl = sqrtf(vdot(dd, dd));
    dd = vdivs(dd, dl);
    ray3 rb = spawn_ray_to(si, lt);
    C->shadow++;
    if (ray_test(sc, &rb, use_bvh)) return;                                   /* :528 */
    v3 wo = to_local(si, dd);
    float bs[3]; bsdf_eval(mat, si->wi, wo, bs);                              /* :531-533 */
    rb.maxt = INFINITY;
    hit_t h2 = intersect(sc, &rb, use_bvh); C->closest++;                      /* :535-537 */
    sinter s2 = make_si(sc, &rb, h2);
    if (!s2.valid) return;
    if (!(bs[0] > ORC_EPS || bs[1] > ORC_EPS || bs[2] > ORC_EPS)) return;     /* :539-540 */
    v3 wl = to_local(&s2, vneg(dd));
    if (!(wl.z > 0.0f)) return;                                               /* :543 */
    float pdf_ls = (dl * dl) / wl.z;                                          /* :546-551 */
    float b2[3];
    for (int k = 0; k < 3; ++k) b2[k] = beta[k] * (bs[k] / pdf_ls);
    mtr_material m2copy;
    const mtr_material *m2 = material_at(sc->d, &sc->d->materials[sc->d->tri_material[s2.prim]], &s2, &m2copy);
    nlos_emitter_nee(sc, N, F, rng, &s2, m2, b2, distance + dl * eta, eta, depth + 1, 1, 1, px, py, sample_scale, lane,
                     loop_depth, use_bvh, C, Lr, laser_x, laser_y);
}

/* hidden_geometry_sample (transientnlospath.py:637-670) */
static void nlos_hidden_geometry(const orc_scene *sc, const nlos_scene *N, const sinter *si, const mtr_material *mat,
                                 float ua, float ub, int active, bsample *bs)
{
    const mtr_scene_desc *d = sc->d; const mtr_nlos_desc *n = N->n;
    memset(bs, 0, sizeof *bs);                      /* dr.zeros(BSDFSample3f): eta 0 ... */
    bs->eta = 1.0f;                                 /* ... :662 bs.eta = 1.0 */
    if (!active) return;
    /* _sample_hidden_geometry_position (:385-430) */
    float reused, spmf;
    uint32_t s = distr_sample_reuse(N->shape_cdf, N->shape_pmf, n->n_shapes, ua, &reused, &spmf);
    const mtr_shape *S = &n->shapes[s];
    v3 pp, pn; float ppdf;
    if (S->is_rectangle) {
        pp = rect_point(V(S->center[0], S->center[1], S->center[2]), V(S->du[0], S->du[1], S->du[2]),
                        V(S->dv[0], S->dv[1], S->dv[2]), reused, ub);
        pn = N->rect_n[s];
    } else {                                        /* [mitsuba3: Mesh::sample_position] */
        float r2, fpmf, sy = ub;
        uint32_t fi = 0;
        if (S->n_tris > 1) fi = distr_sample_reuse(N->face_cdf + S->first_tri, N->face_pmf + S->first_tri, S->n_tris, ub, &r2, &fpmf), sy = r2;
        const orc_tri *T = &sc->tris[S->first_tri + fi];
        float t = sqrtf(fmaxf(1.0f - reused, 0.0f));           /* warp::square_to_uniform_triangle */
        float b0 = 1.0f - t, b1 = t * sy;
        pp = V(fmaf(T->e1.x, b0, fmaf(T->e2.x, b1, T->p0.x)), fmaf(T->e1.y, b0, fmaf(T->e2.y, b1, T->p0.y)),
               fmaf(T->e1.z, b0, fmaf(T->e2.z, b1, T->p0.z)));
        pn = T->n;
        if (T->smooth) {                            /* has_vertex_normals: ps.n = normalize(fmadd(n0, 1 - b.x - b.y, fmadd(n1, b.x, n2 * b.y))) */
            const float *vn = d->tri_normals + 9 * (size_t)(S->first_tri + fi);
            float w0 = 1.0f - b0 - b1;
            pn = vnormalize(V(fmaf(vn[0], w0, fmaf(vn[3], b0, vn[6] * b1)), fmaf(vn[1], w0, fmaf(vn[4], b0, vn[7] * b1)),
                              fmaf(vn[2], w0, fmaf(vn[5], b0, vn[8] * b1))));
        }
    }
    ppdf = N->shape_inv_area[s] * spmf;
    v3 dd = vsub(pp, si->p);
    float dist = sqrtf(vdot(dd, dd));
    dd = vdivs(dd, dist);
    float cos_i = vdot(si->ng, dd), cos_g = vdot(pn, vneg(dd));          /* si.n: the geometric normal */
    if (!(cos_i > ORC_EPS && cos_g > ORC_EPS)) { bs->wo = to_local(si, dd); bs->pdf = ppdf * (dist * dist) / fabsf(cos_g); return; }
    v3 wo = to_local(si, dd);
    float val[3]; bsdf_eval(mat, si->wi, wo, val);
    bs->wo = wo;
    bs->pdf = ppdf * (dist * dist) / fabsf(cos_g);
    if (!(bs->pdf > ORC_EPS)) return;
    for (int k = 0; k < 3; ++k) bs->w[k] = val[k] / bs->pdf;
}

/* one lane of TransientNLOSPath.sample (transientnlospath.py:672-927) */
static void trace_lane_nlos(const orc_scene *sc, const nlos_scene *N, const mtr_render_params *P, film_t *F, uint32_t lane,
                            int use_bvh, lane_counters *C)
{
    const mtr_scene_desc *d = sc->d; const mtr_film_desc *f = &d->film; const mtr_nlos_desc *n = N->n;
    const uint32_t spp = P->spp_total;
    const float sample_scale = (float)(1.0 / (double)(P->spp_scale ? P->spp_scale : spp));
    const uint32_t max_depth = P->max_depth < 0 ? 0xffffffffu : (uint32_t)P->max_depth;
    const uint32_t rr_depth = (uint32_t)P->rr_depth;
    uint32_t idx = lane / spp;
    uint32_t py = idx / f->crop_width, px = idx - f->crop_width * py;
    px += f->crop_offset_x; py += f->crop_offset_y;
    pcg32 rng; sampler_seed_v(&rng, P->seed, lane, P->flags);
    float j1 = pcg32_next_f32(&rng), j2 = pcg32_next_f32(&rng);
    /* sample_rays: pos_adjusted = (pos + jitter) * (1/crop) + offset, then the sensor snaps it to the pixel centre */
    float scx = 1.0f / (float)f->crop_width, scy = 1.0f / (float)f->crop_height;
    float sx = fmaf((float)px + j1, scx, -(float)f->crop_offset_x * scx), sy = fmaf((float)py + j2, scy, -(float)f->crop_offset_y * scy);
    ray3 ray = n->relay_shape == MTR_NLOS_NO_RELAY ? sample_ray(d, px, py, j1, j2) : nlos_sensor_ray(N, f, sx, sy);

    uint32_t depth = 0; float L[3] = { 0, 0, 0 }, beta[3] = { 1, 1, 1 };
    float eta = 1.0f, distance = 0.0f;                                  /* distance = ray.time = 0 (:718) */
    int active = 1;
    /* confocal: this pixel's illuminated point = its scanned point (:337-339, :585-589) */
    uint32_t fx = px - f->crop_offset_x, fy = py - f->crop_offset_y;
    v3 lt = N->laser_target_single;
    if (n->capture_type == MTR_CAPTURE_CONFOCAL) lt = N->sensor_targets[(size_t)py * f->width + px];
    (void)fx; (void)fy;

    while (active) {
        C->bounces++;
        int active_next = 1;
        hit_t h = intersect(sc, &ray, use_bvh); C->closest++;
        sinter si = make_si(sc, &ray, h);
        if ((n->flags & MTR_NLOS_ACCOUNT_FIRST_LAST) || depth > 0) distance += si.t * eta;       /* :751-752 */
        mtr_material matcopy;
        const mtr_material *mat = si.valid ? material_at(d, &d->materials[d->tri_material[si.prim]], &si, &matcopy) : NULL;
        /* direct emission (:757-777): the only emitter is the projector, which is not a surface: Le = 0 */
        active_next &= (depth + 1 < max_depth) && si.valid;                                       /* :782 */
        int active_em = active_next && bsdf_is_smooth(mat);
        float Lr[3] = { 0, 0, 0 };
        if ((n->flags & MTR_NLOS_LASER_SAMPLING) && n->capture_type == MTR_CAPTURE_EXHAUSTIVE) {
            /* each measured point is illuminated by all the laser points (:590-621) */
            const uint32_t lrx = N->laser_w, lry = N->laser_h;
            for (uint32_t i = 0; i < lrx * lry; ++i) {
                uint32_t laser_x = (uint32_t)((float)i / (float)lry), laser_y = i % lry;          /* :609-610 */
                float Li[3];
                nlos_laser_targets(sc, N, F, &rng, &si, mat, N->laser_targets[i], beta, distance, eta, depth + 1, active_em, px, py,
                                   sample_scale, lane, depth, use_bvh, C, Li, laser_x, laser_y);
                for (int k = 0; k < 3; ++k) Lr[k] += Li[k];
            }
            for (int k = 0; k < 3; ++k) Lr[k] = Lr[k] / (float)(lrx * lry);                       /* :621 */
        } else if (n->flags & MTR_NLOS_LASER_SAMPLING)                                            /* emitter_laser_sample: depth + 1 */
            nlos_laser_targets(sc, N, F, &rng, &si, mat, lt, beta, distance, eta, depth + 1, active_em, px, py, sample_scale,
                               lane, depth, use_bvh, C, Lr, 0, 0);
        else
            nlos_emitter_nee(sc, N, F, &rng, &si, mat, beta, distance, eta, depth, active_em, 0, px, py, sample_scale, lane,
                             depth, use_bvh, C, Lr, 0, 0);
        /* BSDF / hidden-geometry sampling (:797-833) */
        int do_hg = (n->flags & MTR_NLOS_HG_SAMPLING) != 0; float pdf_method = 1.0f;
        if ((n->flags & MTR_NLOS_HG_SAMPLING) && (n->flags & MTR_NLOS_HG_RROULETTE)) {
            float u = pcg32_next_f32(&rng);                                                     /* next_1d(active) :801 */
            do_hg = u < 0.5f; pdf_method = 0.5f;
        }
        float a1 = pcg32_next_f32(&rng), a2a = pcg32_next_f32(&rng), a2b = pcg32_next_f32(&rng); (void)a1;   /* :814 */
        bsample bs_hg; nlos_hidden_geometry(sc, N, &si, mat, a2a, a2b, active_next && do_hg && (n->flags & MTR_NLOS_HG_SAMPLING), &bs_hg);
        if (!(n->flags & MTR_NLOS_HG_SAMPLING)) { memset(&bs_hg, 0, sizeof bs_hg); }                       /* zeros(BSDFSample3f), 0 */
        float b1 = pcg32_next_f32(&rng), b2a = pcg32_next_f32(&rng), b2b = pcg32_next_f32(&rng);          /* :820 */
        bsample bs_n; memset(&bs_n, 0, sizeof bs_n); bs_n.eta = 1.0f;
        if (active_next && !do_hg) bsdf_sample(mat, si.wi, b1, b2a, b2b, &bs_n);
        bsample bs = do_hg ? bs_hg : bs_n;
        for (int k = 0; k < 3; ++k) L[k] = L[k] + Lr[k];
        if (active_next) {
            v3 wo_w = to_world(&si, bs.wo);
            ray.o = offset_p(&si, wo_w); ray.d = wo_w; ray.maxt = INFINITY;
        }
        eta *= bs.eta;
        for (int k = 0; k < 3; ++k) beta[k] = (beta[k] * bs.w[k]) / pdf_method;                  /* :833 */
        float bmax = fmaxf(beta[0], fmaxf(beta[1], beta[2]));
        active_next &= (bmax != 0.0f);
        float rr_prob = fminf(bmax * (eta * eta), 0.95f);
        active_next &= rr_prob > 0.0f;
        int rr_active = depth >= rr_depth;
        if (rr_active) { float inv = rr_prob > 0.0f ? 1.0f / rr_prob : 0.0f; for (int k = 0; k < 3; ++k) beta[k] *= inv; }
        float rr_u = pcg32_next_f32(&rng);
        active_next &= (!rr_active) || (rr_u < rr_prob);
        if (si.valid) depth += 1;
        active = active_next;
    }
    if (F->steady) {
        uint32_t x = px - f->crop_offset_x, y = py - f->crop_offset_y;
        if (x < f->width && y < f->height) {
            size_t i = ((size_t)y * f->width + x) * 4u;
            for (int k = 0; k < 3; ++k) {
#pragma omp atomic
                F->steady[i + k] += L[k];
            }
#pragma omp atomic
            F->steady[i + 3] += 1.0f;
        }
    }
}

/* ------------------------------------------------------------------ */
/* Public entry points                                                 */
/* ------------------------------------------------------------------ */
static int g_default_threads = 0;
int orc_render(const mtr_scene_desc *d, const mtr_render_params *P, float *transient_hwt4, float *steady_hw4,
               mtr_counters *out, int n_threads, int use_bvh,
               orc_splat_rec *log, uint64_t log_cap, uint64_t *log_n)
{
    if (!d || !P || !transient_hwt4) return -1;
    if (P->spp_total == 0 || P->spp_end > P->spp_total || P->spp_begin > P->spp_end) return -1;
    if ((uint64_t)d->film.crop_width * d->film.crop_height * P->spp_total > (1ull << 32)) return -2;   /* common.py:51 */
    if (P->pixel_end > d->film.crop_width * d->film.crop_height || P->pixel_begin > P->pixel_end) return -1;
    orc_scene sc; memset(&sc, 0, sizeof sc); sc.d = d;
    build_tris(&sc);
    if (use_bvh) build_bvh(&sc);
    film_t F; memset(&F, 0, sizeof F);
    F.f = &d->film; F.transient = transient_hwt4; F.steady = steady_hw4;
    uint64_t zero = 0; F.log = log; F.log_cap = log_cap; F.log_n = log_n ? log_n : &zero;
    if (log_n) *log_n = 0;
    nlos_scene NS; memset(&NS, 0, sizeof NS);
    if (d->nlos) nlos_build(&NS, &sc, use_bvh);
    uint64_t closest = 0, shadow = 0, bounces = 0, paths = 0, splats = 0;
    const int64_t n_pix = (int64_t)P->pixel_end - (int64_t)P->pixel_begin;
    const uint32_t s0 = P->spp_begin, s1 = P->spp_end;
#ifdef _OPENMP
    /* n_threads <= 0: the process's default team (OMP_NUM_THREADS or every core) — also after an earlier call asked for fewer
     * (omp_set_num_threads persists: a test that rendered on one thread left every later render on one thread) */
    if (!g_default_threads) g_default_threads = omp_get_max_threads();
    omp_set_num_threads(n_threads > 0 ? n_threads : g_default_threads);
#else
    (void)n_threads;
#endif
#pragma omp parallel for schedule(dynamic, 16) reduction(+ : closest, shadow, bounces, paths, splats)
    for (int64_t ip = 0; ip < n_pix; ++ip) {
        uint32_t pix = P->pixel_begin + (uint32_t)ip;
        for (uint32_t s = s0; s < s1; ++s) {
            uint32_t lane = pix * P->spp_total + s;      /* lane identity == RNG identity */
            lane_counters C = { 0, 0, 0, 0 };
            if (d->nlos) trace_lane_nlos(&sc, &NS, P, &F, lane, use_bvh, &C);
            else trace_lane(&sc, P, &F, lane, use_bvh, &C);
            closest += C.closest; shadow += C.shadow; bounces += C.bounces; splats += C.splats; paths += 1;
        }
    }
    if (out) {
        memset(out, 0, sizeof *out);
        out->paths = paths; out->rays_closest = closest; out->rays_shadow = shadow;
        out->bounces = bounces; out->splats_issued = splats;
    }
    if (d->nlos) nlos_free(&NS);
    free_scene(&sc);
    return 0;
}

/* develop (transient_hdr_film.py:220-248; steady hdrfilm: sum / weight) */
void orc_develop(const mtr_film_desc *f, const float *transient_hwt4, float *transient_hwt3,
                 const float *steady_hw4, float *steady_hw3)
{
    size_t npt = film_cells(f);
    if (f->n_frequencies) {                        /* develop_phasors_ (phasor_hdr_film.py:216-238): (H,W,2F+1) -> (H,W,F,2) */
        const size_t np_ = (size_t)f->width * f->height, F2 = 2u * (size_t)f->n_frequencies;
        if (transient_hwt4 && transient_hwt3)
            for (size_t i = 0; i < np_; ++i) {
                float w = transient_hwt4[i * (F2 + 1) + F2];
                float dv = (w == 0.0f) ? 1.0f : w;
                for (size_t k = 0; k < F2; ++k) transient_hwt3[i * F2 + k] = transient_hwt4[i * (F2 + 1) + k] / dv;
            }
        transient_hwt4 = NULL;
    }
    if (transient_hwt4 && transient_hwt3)
        for (size_t i = 0; i < npt; ++i) {
            float w = transient_hwt4[4 * i + 3];
            float dv = (w == 0.0f) ? 1.0f : w;
            for (int k = 0; k < 3; ++k) transient_hwt3[3 * i + k] = transient_hwt4[4 * i + k] / dv;
        }
    size_t np = (size_t)f->width * f->height;
    if (steady_hw4 && steady_hw3)
        for (size_t i = 0; i < np; ++i) {
            float w = steady_hw4[4 * i + 3];
            for (int k = 0; k < 3; ++k) steady_hw3[3 * i + k] = (w != 0.0f) ? steady_hw4[4 * i + k] / w : 0.0f;
        }
}

/* stand-alone splat add (same arithmetic as add_transient) for the scatter-add tests */
void orc_splat_add(const mtr_film_desc *f, uint64_t n, const uint32_t *pixel, const float *opl,
                   const float *r, const float *g, const float *b, float *transient_hwt4,
                   const uint32_t *laser_x, const uint32_t *laser_y)
{
    for (uint64_t i = 0; i < n; ++i) {
        if (f->n_frequencies) {
            if (!isfinite(opl[i]) || pixel[i] >= f->width * f->height) continue;
            float *dst = transient_hwt4 + (size_t)pixel[i] * (2u * (size_t)f->n_frequencies + 1u);
            for (uint32_t k = 0; k < f->n_frequencies; ++k) {
                float c, sn;
                phasor_term(f->frequencies[k], opl[i] - f->start_opl, &c, &sn);
                dst[2 * k] += r[i] * c; dst[2 * k + 1] += r[i] * sn;
            }
            continue;
        }
        int bin = orc_bin_index(opl[i], f->start_opl, f->bin_width_opl, f->temporal_bins);
        if (bin < 0) continue;
        if (pixel[i] >= f->width * f->height) continue;
        uint32_t lx = laser_x ? laser_x[i] : 0, ly = laser_y ? laser_y[i] : 0;
        if (f->laser_scan_width && !(lx < f->laser_scan_width && ly < f->laser_scan_height)) continue;
        size_t index = film_cell(f, pixel[i] % f->width, pixel[i] / f->width, lx, ly, (uint32_t)bin) * 4u;
        transient_hwt4[index + 0] += r[i]; transient_hwt4[index + 1] += g[i]; transient_hwt4[index + 2] += b[i];
    }
}

/* closest-hit probe for BVH/intersection tests */
void orc_intersect(const mtr_scene_desc *d, uint32_t n, const float *o3, const float *d3, const float *maxt,
                   int use_bvh, float *t_out, int32_t *prim_out, uint8_t *occluded_out)
{
    orc_scene sc; memset(&sc, 0, sizeof sc); sc.d = d;
    build_tris(&sc);
    if (use_bvh) build_bvh(&sc);
    for (uint32_t i = 0; i < n; ++i) {
        ray3 r; r.o = V(o3[3 * i], o3[3 * i + 1], o3[3 * i + 2]); r.d = V(d3[3 * i], d3[3 * i + 1], d3[3 * i + 2]);
        r.maxt = maxt ? maxt[i] : INFINITY;
        hit_t h = intersect(&sc, &r, use_bvh);
        if (t_out) t_out[i] = h.t;
        if (prim_out) prim_out[i] = h.prim;
        if (occluded_out) occluded_out[i] = (uint8_t)ray_test(&sc, &r, use_bvh);
    }
    free_scene(&sc);
}

void orc_camera_ray(const mtr_scene_desc *d, uint32_t px, uint32_t py, float j1, float j2, float *o3, float *d3, float *maxt)
{
    ray3 r = sample_ray(d, px, py, j1, j2);
    o3[0] = r.o.x; o3[1] = r.o.y; o3[2] = r.o.z; d3[0] = r.d.x; d3[1] = r.d.y; d3[2] = r.d.z; *maxt = r.maxt;
}

int orc_num_threads(void)
{
#ifdef _OPENMP
    return g_default_threads ? g_default_threads : omp_get_max_threads();
#else
    return 1;
#endif
}
