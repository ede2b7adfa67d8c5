#include "mtr_scene_host.h"
#include <chrono>
#include <cstdio>
#include "mtr_knobs.h"
#include <cstring>
#include <cmath>

namespace mtr {

Film film_from_desc(const mtr_film_desc &d)
{
    Film f;
    f.width = d.width; f.height = d.height; f.crop_w = d.crop_width; f.crop_h = d.crop_height;
    f.crop_x = d.crop_offset_x; f.crop_y = d.crop_offset_y;
    f.tbins = d.temporal_bins;
    f.lasers = (d.laser_scan_width && d.laser_scan_height) ? d.laser_scan_width * d.laser_scan_height : 1u;
    f.bins = f.tbins * f.lasers;
    f.start_opl = d.start_opl; f.bin_width = d.bin_width_opl;
    f.n_freq = d.n_frequencies; f.freq = d.frequencies;         // host pointer; the API layer swaps in its device copy
    if (f.n_freq) { f.tbins = 1; f.lasers = 1; f.bins = 1; }
    return f;
}

static double tri_area_d(const float *v)
{
    double e1[3] = { (double)v[3] - v[0], (double)v[4] - v[1], (double)v[5] - v[2] };
    double e2[3] = { (double)v[6] - v[0], (double)v[7] - v[1], (double)v[8] - v[2] };
    double c[3] = { e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0] };
    return 0.5 * sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
}

// area-sampling tables of one mesh (triangles [first, first + n) of the ORIGINAL order): face distribution with f64
// sums stored as f32 (Mesh::build_pmf), and (p0, e1, e2, n) quads; returns the mesh area
static double fill_mesh_tables(const float *tri_verts, uint32_t first, uint32_t n, float *pmf, float *cdf, q4 *quads)
{
    double a = 0.0, acc = 0.0;
    for (uint32_t t = 0; t < n; ++t) a += tri_area_d(tri_verts + 9 * (size_t)(first + t));
    for (uint32_t t = 0; t < n; ++t) {
        const float *v = tri_verts + 9 * (size_t)(first + t);
        const double at = tri_area_d(v);
        acc += at;
        pmf[first + t] = (float)(at / a);
        cdf[first + t] = (float)(acc / a);
        const f3 p0 = mk(v[0], v[1], v[2]), e1 = mk(v[3], v[4], v[5]) - p0, e2 = mk(v[6], v[7], v[8]) - p0;
        const f3 nn = normalize(cross(e1, e2));
        q4 *q = &quads[3 * (size_t)(first + t)];
        q[0] = q4{ p0.x, p0.y, p0.z, e1.x }; q[1] = q4{ e1.y, e1.z, e2.x, e2.y }; q[2] = q4{ e2.z, nn.x, nn.y, nn.z };
    }
    return a;
}

// vertex normals of the same triangles for Mesh::sample_position (has_vertex_normals): three quads per ORIGINAL triangle,
// .w of the first = 1 where the scene holds normals for it; `quads` is sized (3 * n_total) on first use
static void fill_mesh_normals(const float *tri_normals, uint32_t first, uint32_t n, uint32_t n_total, std::vector<q4> &quads)
{
    if (!tri_normals) return;
    for (uint32_t t = 0; t < n; ++t) {
        const float *vn = tri_normals + 9 * (size_t)(first + t);
        bool smooth = false;
        for (int k = 0; k < 9; ++k) smooth = smooth || vn[k] != 0.0f;
        if (!smooth) continue;
        if (quads.empty()) quads.assign(3 * (size_t)n_total, q4{ 0, 0, 0, 0 });
        for (int k = 0; k < 3; ++k) quads[3 * (size_t)(first + t) + k] = q4{ vn[3 * k], vn[3 * k + 1], vn[3 * k + 2], k == 0 ? 1.0f : 0.0f };
    }
}

// [mitsuba3: coordinate_system(n)] (Duff et al. 2017), first vector: the tangent of a mesh triangle without UVs
static f3 coordinate_system_s(f3 n)
{
    const float sign = copysignf(1.0f, n.z);
    const float a = -(1.0f / (sign + n.z));
    const float b = (n.x * n.y) * a;
    const bool neg = sign_neg(n.z);
    const float x = (n.x * n.x) * a, nx = -n.x;
    return mk((neg ? -x : x) + 1.0f, neg ? -b : b, neg ? -nx : nx);
}
// [mitsuba3: SurfaceInteraction::initialize_sh_frame]
static void sh_frame_from(f3 n, f3 dp_du, f3 &s, f3 &t) { sh_frame_of(n, dp_du, s, t); }
// rows of a rectangle's to_object from (c, du, dv): the inverse of [du dv n^ | c], n^ = normalize(du x dv); f64 -> f32
// (numerics contract: the same operations, in the same order, as the test oracle's restatement)
static void rect_to_object(const float c[3], const float du[3], const float dv[3], float rx[4], float ry[4], float rz[4])
{
    const double a[3] = { du[0], du[1], du[2] }, b[3] = { dv[0], dv[1], dv[2] }, o[3] = { c[0], c[1], c[2] };
    double n[3] = { a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0] };
    const double ln = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    n[0] /= ln; n[1] /= ln; n[2] /= ln;
    const double bn[3] = { b[1] * n[2] - b[2] * n[1], b[2] * n[0] - b[0] * n[2], b[0] * n[1] - b[1] * n[0] };
    const double na[3] = { n[1] * a[2] - n[2] * a[1], n[2] * a[0] - n[0] * a[2], n[0] * a[1] - n[1] * a[0] };
    const double da = a[0] * bn[0] + a[1] * bn[1] + a[2] * bn[2], db = b[0] * na[0] + b[1] * na[1] + b[2] * na[2];
    const double X[3] = { bn[0] / da, bn[1] / da, bn[2] / da }, Y[3] = { na[0] / db, na[1] / db, na[2] / db };
    for (int k = 0; k < 3; ++k) { rx[k] = (float)X[k]; ry[k] = (float)Y[k]; rz[k] = (float)n[k]; }
    rx[3] = (float)(-(X[0] * o[0] + X[1] * o[1] + X[2] * o[2]));
    ry[3] = (float)(-(Y[0] * o[0] + Y[1] * o[1] + Y[2] * o[2]));
    rz[3] = (float)(-(n[0] * o[0] + n[1] * o[1] + n[2] * o[2]));
}
// world -> object rows (R | T) of a mesh shape's 3 x 4 to_world; false when the map is singular or axis-aligned (every
// row of the linear part has one entry: object-space boxes would be the world boxes)
static bool object_space_of(const float tw[12], float inv[12], bool *axis_aligned = nullptr)
{
    const double m[9] = { tw[0], tw[1], tw[2], tw[4], tw[5], tw[6], tw[8], tw[9], tw[10] }, t[3] = { tw[3], tw[7], tw[11] };
    double big = 0.0;
    for (double x : m) big = fmax(big, fabs(x));
    bool aligned = true;
    for (int r = 0; r < 3; ++r) {
        int nz = 0;
        for (int c = 0; c < 3; ++c) nz += fabs(m[3 * r + c]) > 1e-6 * big ? 1 : 0;
        aligned = aligned && nz <= 1;
    }
    const double det = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
    if (axis_aligned) *axis_aligned = aligned;
    if ((aligned && !axis_aligned) || !(fabs(det) > 1e-30)) return false;
    const double c[9] = { m[4] * m[8] - m[5] * m[7], m[2] * m[7] - m[1] * m[8], m[1] * m[5] - m[2] * m[4],
                          m[5] * m[6] - m[3] * m[8], m[0] * m[8] - m[2] * m[6], m[2] * m[3] - m[0] * m[5],
                          m[3] * m[7] - m[4] * m[6], m[1] * m[6] - m[0] * m[7], m[0] * m[4] - m[1] * m[3] };
    for (int r = 0; r < 3; ++r) {
        double row[3] = { c[3 * r] / det, c[3 * r + 1] / det, c[3 * r + 2] / det };
        for (int k = 0; k < 3; ++k) inv[4 * r + k] = (float)row[k];
        inv[4 * r + 3] = (float)(-(row[0] * t[0] + row[1] * t[1] + row[2] * t[2]));
    }
    return true;
}

const char *derive_scene(const mtr_scene_desc &d, HostScene &s)
{
    if (d.n_tris && (!d.tri_verts || !d.tri_material || !d.tri_emitter)) return "triangle arrays missing";
    if (d.n_materials > 0xffffu || d.n_emitters > 0x7fffu) return "too many materials/emitters";
    for (uint32_t i = 0; i < d.n_tris; ++i) {
        if (d.tri_material[i] >= d.n_materials) return "triangle references an unknown material";
        if (d.tri_emitter[i] >= (int32_t)d.n_emitters) return "triangle references an unknown emitter";
    }
    for (uint32_t i = 0; i < d.n_materials; ++i)
        if (d.materials[i].type > MTR_BSDF_PLASTIC) return "unknown BSDF type";
    for (uint32_t i = 0; i < d.n_materials; ++i) {
        if (bsdf_is_rough(d.materials[i].type) && d.materials[i].type != MTR_BSDF_PLASTIC && !(d.materials[i].alpha > 0.0f)) return "rough BSDF: alpha must be positive";
        // the second roughness of an anisotropic lobe (c2[0]; roughdielectric: b[0]) divides in ggx_eval / beck_eval just like alpha
        if (bsdf_is_rough(d.materials[i].type) && d.materials[i].type != MTR_BSDF_PLASTIC && (d.materials[i].flags & MTR_MAT_ANISOTROPIC) &&
            !(rough_alpha_v(d.materials[i]) > 0.0f)) return "rough BSDF: alpha_v of an anisotropic lobe must be positive";
        // the two-sided adapter is defined for materials without a transmission component only (the header's contract; scene.py enforces it too)
        if ((d.materials[i].flags & MTR_MAT_TWOSIDED) && (d.materials[i].type == MTR_BSDF_DIELECTRIC || d.materials[i].type == MTR_BSDF_ROUGHDIELECTRIC ||
                                                           d.materials[i].type == MTR_BSDF_THINDIELECTRIC)) return "MTR_MAT_TWOSIDED on a transmissive BSDF";
    }

    s.film = film_from_desc(d.film);
    memcpy(s.cam.s2c, d.camera.sample_to_camera, sizeof s.cam.s2c);
    memcpy(s.cam.tw, d.camera.to_world, sizeof s.cam.tw);
    s.cam.near_clip = d.camera.near_clip; s.cam.far_clip = d.camera.far_clip;

    // shape table -> per-triangle annotations of the BVH build: analytic rectangles, and OBJECTS = small mesh shapes
    // with a known, not axis-aligned object -> world transform (a rotated `cube`): their triangles get object-space
    // bounds in the 8-wide tree (mtr_core.h, WNodeT)
    std::vector<uint8_t> kind(d.n_tris, 0);
    std::vector<int32_t> object(d.n_tris, -1);
    std::vector<float> object_xf;
    std::vector<const mtr_shape *> rect_of(d.n_tris, nullptr);
    uint32_t covered = 0;
    for (uint32_t k = 0; k < d.n_shapes && d.shapes; ++k) {
        const mtr_shape &S = d.shapes[k];
        if (S.first_tri != covered || (uint64_t)S.first_tri + S.n_tris > d.n_tris) return "shapes must tile the triangle array in order";
        covered += S.n_tris;
        if (S.is_rectangle) {
            if (S.n_tris != 2) return "a rectangle shape owns exactly two carrier triangles";
            kind[S.first_tri] = 1; kind[S.first_tri + 1] = 2;
            rect_of[S.first_tri] = rect_of[S.first_tri + 1] = &S;
            continue;
        }
        float inv[12];
        bool aligned = false;
        uint32_t faces[12];
        // (an axis-aligned transform gains nothing from object-space boxes — unless the shape is an affine cube, which
        // becomes a box node whatever its orientation)
        if (S.has_to_world && S.n_tris >= 4 && S.n_tris <= 16 && object_space_of(S.to_world, inv, &aligned) &&
            (!aligned || (S.n_tris == 12 && mesh_is_affine_box(d.tri_verts, S.first_tri, inv, faces)))) {
            for (uint32_t t = 0; t < S.n_tris; ++t) object[S.first_tri + t] = (int32_t)(object_xf.size() / 12);
            object_xf.insert(object_xf.end(), inv, inv + 12);
        }
    }
    if (d.n_shapes && d.shapes && covered != d.n_tris) return "shapes must cover every triangle";
    BvhPrims prims;
    prims.kind = kind.data(); prims.object = object.data(); prims.object_xf = object_xf.empty() ? nullptr : object_xf.data();

    // BVH2 over the primitives; triangles are stored in leaf order
    BvhBuild bvh;
    const bool verbose = mtr::knob("MTR_BVH_VERBOSE") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_phase = now();
    auto phase = [&](const char *what) { if (verbose) { const double t = now(); fprintf(stderr, "derive_scene: %-28s %.3f s\n", what, t - t_phase); t_phase = t; } };
    build_bvh(d.tri_verts, d.n_tris, &prims, bvh);
    phase("build_bvh");
    s.nodes = bvh.nodes; s.bvh_depth = bvh.max_depth; s.n_leaves = bvh.n_leaves;
    s.wnodes.clear();
    s.has_wide = bvh.nodes.size() <= 2048;
    s.wide_levels = s.has_wide ? build_wide(bvh, &prims, d.tri_verts, s.wnodes) : 0;
    s.wnodes4.clear();
    s.wide4_levels = build_wide4(bvh, s.wnodes4);
    phase("build_wide + build_wide4");
    s.wnodes8q.clear(); s.wide8q_levels = 0;
    if (!mtr::knob("MTR_NO_WIDE8Q")) s.wide8q_levels = build_wide8q(bvh, s.wnodes8q);      // (experiments: without it the HBM walk uses the 4-wide tree)
    phase("build_wide8q");
    const uint32_t n_slots = (uint32_t)bvh.order.size();
    s.tpairs.assign(n_slots / 2, TriPair{}); s.tshade.assign(n_slots, TriShade{}); s.slot_orig.assign(n_slots, 0u);
    s.vnormals.clear(); s.samp_vn.clear();
    // bitmap textures: texels to RGBA f32, texture coordinates by slot (filled in the slot loop below)
    s.texels.clear(); s.tex_info.clear(); s.uvs.clear();
    bool textured = false;
    for (uint32_t i = 0; i < d.n_materials; ++i) {
        if (d.materials[i].albedo_texture > d.n_textures) return "material references an unknown texture";
        textured = textured || d.materials[i].albedo_texture != 0u;
    }
    if (textured) {
        if (!d.textures) return "textures missing";
        for (uint32_t t = 0; t < d.n_textures; ++t) {
            const mtr_texture &T = d.textures[t];
            if (!T.rgb || T.width == 0 || T.height == 0 || T.width > 16384u || T.height > 16384u) return "bad texture";
            s.tex_info.push_back(q4{ bitsf((uint32_t)s.texels.size()), bitsf(T.width), bitsf(T.height), 0.0f });
            for (size_t k = 0; k < (size_t)T.width * T.height; ++k) s.texels.push_back(q4{ T.rgb[3 * k], T.rgb[3 * k + 1], T.rgb[3 * k + 2], 0.0f });
        }
        s.uvs.assign(2 * (size_t)n_slots, q4{ 0, 0, 0, 0 });
    }
    if (d.n_tris >= (1u << 29)) return "too many triangles (the tie-break word holds 29 bits of triangle index)";
    // (a leaf's code is (first slot << 2 | count - 1) under kLeafQuadBit = bit 30; spatial splits duplicate references: up to 2 slots per triangle)
    if (n_slots >= (1u << 28)) return "too many triangle slots (a leaf code holds 28 bits of slot index)";
    // the tie-break word of an intersection record: (original index << 3) | list key of the triangle's material (mtr_core.h hit_list_key)
    auto tie_word = [&](uint32_t orig) { return (orig << 3) | hit_list_key(d.materials[d.tri_material[orig]].type); };
    for (uint32_t slot = 0; slot < n_slots; ++slot) {
        const uint32_t o = bvh.order[slot] != kPadSlot ? bvh.order[slot] : bvh.order[slot - 1];   // pad: repeat the leaf's last triangle
        s.slot_orig[slot] = o;
        const float *v = d.tri_verts + 9 * (size_t)o;
        TriShade &h = s.tshade[slot];
        const uint32_t mat_em = d.tri_material[o] | ((uint32_t)(d.tri_emitter[o] + 1) << 16);
        if (const mtr_shape *R = rect_of[o]) {
            // analytic rectangle [mitsuba3: src/shapes/rectangle.cpp]: both slots of its pair describe the ONE primitive
            // (index = its first carrier triangle)
            const uint32_t prim = R->first_tri;
            const f3 c = ld3(R->center), du = ld3(R->du), dv = ld3(R->dv);
            f3 n = normalize(cross(du, dv));                                    // normalize(to_world * Normal3f(0, 0, 1))
            if (R->is_rectangle & MTR_RECT_FLIP_NORMALS) n = mk(-n.x, -n.y, -n.z);   // [Rectangle: flip_normals negates the frame normal, not the parameterisation]
            f3 sdir, t;
            sh_frame_from(n, du, sdir, t);                                      // dp_du = to_world * (2, 0, 0): same direction
            float rx[4], ry[4], rz[4];
            rect_to_object(R->center, R->du, R->dv, rx, ry, rz);
            TriPair &tp = s.tpairs[slot >> 1];
            tp.g[0] = q4{ rz[0], rz[1], rz[2], rz[3] }; tp.g[1] = q4{ rx[0], rx[1], rx[2], rx[3] }; tp.g[2] = q4{ ry[0], ry[1], ry[2], ry[3] };
            tp.g[3] = q4{ 0, 0, 0, 0 }; tp.g[4] = q4{ 0, 0, bitsf(tie_word(prim)), bitsf(kQuadMark) };
            h.h[0] = q4{ n.x, n.y, n.z, sdir.x };
            h.h[1] = q4{ sdir.y, sdir.z, t.x, t.y };
            h.h[2] = q4{ t.z, du.x, du.y, du.z };
            h.h[3] = q4{ dv.x, dv.y, dv.z, c.x };
            h.h[4] = q4{ c.y, c.z, bitsf(d.tri_material[prim] | ((uint32_t)(d.tri_emitter[prim] + 1) << 16)), bitsf(prim | kShadeQuadBit) };
            continue;
        }
        // flat frame [mitsuba3: Mesh::compute_surface_interaction + SurfaceInteraction::initialize_sh_frame]:
        // n = normalize(e1 x e2); dp_du from the UV parameterisation when there is one and it is not degenerate, else
        // coordinate_system(n); s = normalize(dp_du - n * dot(n, dp_du)), t = n x s   (f32, contract in DESIGN.md)
        f3 p0 = mk(v[0], v[1], v[2]), e1 = mk(v[3], v[4], v[5]) - p0, e2 = mk(v[6], v[7], v[8]) - p0;
        f3 n = normalize(cross(e1, e2));
        f3 dp_du = coordinate_system_s(n);
        if (d.tri_uv) {
            const float *uv = d.tri_uv + 6 * (size_t)o;
            const float duv0x = uv[2] - uv[0], duv0y = uv[3] - uv[1], duv1x = uv[4] - uv[0], duv1y = uv[5] - uv[1];
            const float det = fmaf(duv0x, duv1y, -(duv0y * duv1x));
            if (det != 0.0f) {
                const float inv_det = 1.0f / det;
                dp_du = mk(fmaf(duv1y, e1.x, -(duv0y * e2.x)) * inv_det, fmaf(duv1y, e1.y, -(duv0y * e2.y)) * inv_det,
                           fmaf(duv1y, e1.z, -(duv0y * e2.z)) * inv_det);
            }
        }
        f3 sdir, t;
        sh_frame_from(n, dp_du, sdir, t);
        // smooth-shaded: the frame is built at the hit from the interpolated normal (hit_ctx); the record keeps dp_du
        bool smooth = false;
        if (d.tri_normals) {
            const float *vn = d.tri_normals + 9 * (size_t)o;
            for (int k = 0; k < 9; ++k) smooth = smooth || vn[k] != 0.0f;
            if (smooth) {
                if (s.vnormals.empty()) s.vnormals.assign(3 * (size_t)n_slots, q4{ 0, 0, 0, 0 });
                for (int k = 0; k < 3; ++k) s.vnormals[3 * (size_t)slot + k] = q4{ vn[3 * k], vn[3 * k + 1], vn[3 * k + 2], 0.0f };
                sdir = dp_du; t = mk(0, 0, 0);
            }
        }
        if (textured) {
            // a mesh without texture coordinates (none given, or this shape's are all zero): si.uv = (b1, b2)
            // [Mesh::compute_surface_interaction], i.e. the corners (0,0) (1,0) (0,1)
            const float bary[6] = { 0.0f, 0.0f, 1.0f, 0.0f, 0.0f, 1.0f };
            const float *uv = d.tri_uv ? d.tri_uv + 6 * (size_t)o : bary;
            bool any_uv = false;
            for (int k = 0; k < 6; ++k) any_uv = any_uv || uv[k] != 0.0f;
            if (!any_uv) uv = bary;
            s.uvs[2 * (size_t)slot] = q4{ uv[0], uv[1], uv[2], uv[3] }; s.uvs[2 * (size_t)slot + 1] = q4{ uv[4], uv[5], 0.0f, 0.0f };
        }
        float *g = &s.tpairs[slot >> 1].g[0].x;            // interleaved pair record: dword 2*k + half
        const float comp[9] = { p0.x, p0.y, p0.z, e1.x, e1.y, e1.z, e2.x, e2.y, e2.z };
        const uint32_t half = slot & 1u;
        for (int k = 0; k < 9; ++k) g[2 * k + half] = comp[k];
        g[18 + half] = bitsf(tie_word(o));
        h.h[0] = q4{ n.x, n.y, n.z, sdir.x };
        h.h[1] = q4{ sdir.y, sdir.z, t.x, t.y };
        h.h[2] = q4{ t.z, v[3], v[4], v[5] };
        h.h[3] = q4{ v[6], v[7], v[8], p0.x };
        h.h[4] = q4{ p0.y, p0.z, bitsf(mat_em), bitsf(o | (smooth ? kShadeSmoothBit : 0u)) };
    }
    s.ems.resize(d.n_emitters);
    for (uint32_t i = 0; i < d.n_emitters; ++i) {
        const mtr_emitter &e = d.emitters[i];
        Emitter &E = s.ems[i];
        E.is_mesh = e.is_mesh; E.first_tri = e.first_tri; E.n_tris = e.n_tris; E.pad = 0;
        if (e.is_mesh) {
            if (e.n_tris == 0 || (uint64_t)e.first_tri + e.n_tris > d.n_tris) return "mesh emitter: bad triangle range";
            if (s.samp_tris.empty()) {
                s.samp_tris.assign(3 * (size_t)d.n_tris, q4{ 0, 0, 0, 0 });
                s.face_pmf.assign(d.n_tris, 0.0f); s.face_cdf.assign(d.n_tris, 0.0f);
            }
            const double a = fill_mesh_tables(d.tri_verts, e.first_tri, e.n_tris, s.face_pmf.data(), s.face_cdf.data(), s.samp_tris.data());
            fill_mesh_normals(d.tri_normals, e.first_tri, e.n_tris, d.n_tris, s.samp_vn);
            for (int k = 0; k < 3; ++k) { E.center[k] = E.du[k] = E.dv[k] = E.n[k] = 0.0f; E.radiance[k] = e.radiance[k]; }
            E.inv_area = (float)(1.0 / a);
            continue;
        }
        for (int k = 0; k < 3; ++k) { E.center[k] = e.center[k]; E.du[k] = e.du[k]; E.dv[k] = e.dv[k]; E.radiance[k] = e.radiance[k]; }
        f3 cr = cross(ld3(e.du), ld3(e.dv));
        float len = sqrtf(dot(cr, cr));
        f3 n = cr / len;
        if (e.flip_normals) n = mk(-n.x, -n.y, -n.z);
        E.n[0] = n.x; E.n[1] = n.y; E.n[2] = n.z;
        E.inv_area = 1.0f / (4.0f * len);      // rectangle area = |(2 du) x (2 dv)|
    }
    s.mats.assign(d.materials, d.materials + d.n_materials);
    phase("slots, materials, emitters");
    return nullptr;
}

const char *derive_nlos(const mtr_scene_desc &d, HostNlos &o)
{
    const mtr_nlos_desc *n = d.nlos;
    if (!n) return "no NLOS description";
    if (n->capture_type != MTR_CAPTURE_SINGLE && n->capture_type != MTR_CAPTURE_CONFOCAL &&
        n->capture_type != MTR_CAPTURE_EXHAUSTIVE)
        return "capture_type must be Single, Confocal or Exhaustive";
    const bool exhaustive = n->capture_type == MTR_CAPTURE_EXHAUSTIVE;
    if (exhaustive && !(d.film.laser_scan_width && d.film.laser_scan_height))
        return "Exhaustive capture needs an exhaustive_scan film (laser_scan_width / laser_scan_height > 0)";
    if (exhaustive && (n->flags & MTR_NLOS_FORCE_EQUAL_GRIDS) &&
        (d.film.laser_scan_width != d.film.width || d.film.laser_scan_height != d.film.height))
        return "Sensor and laser scan resolution must be equal if force_equal_illumination_scanning is set to True";
    const bool camera_sensor = n->relay_shape == MTR_NLOS_NO_RELAY;
    if (camera_sensor && (d.film.crop_width != d.film.width || d.film.crop_height != d.film.height))
        return "NLOS with a perspective sensor: crop windows are not supported";
    if (!n->shapes || n->n_shapes == 0 || (!camera_sensor && n->relay_shape >= n->n_shapes)) return "NLOS: bad shape table";
    if (!camera_sensor && !n->shapes[n->relay_shape].is_rectangle) return "NLOS: the relay wall must be a rectangle";
    if (d.n_emitters != 0) return "NLOS: area emitters are not supported next to the projector";
    uint32_t covered = 0;
    for (uint32_t s = 0; s < n->n_shapes; ++s) {
        if (n->shapes[s].first_tri != covered) return "NLOS: shapes must tile the triangle array in order";
        covered += n->shapes[s].n_tris;
    }
    if (covered != d.n_tris) return "NLOS: shapes must cover every triangle";
    NlosConst &k = o.k;
    k.sensor_origin = mk(n->sensor_origin[0], n->sensor_origin[1], n->sensor_origin[2]);
    k.camera_sensor = camera_sensor ? 1u : 0u;
    memcpy(k.cam.s2c, d.camera.sample_to_camera, sizeof k.cam.s2c);
    memcpy(k.cam.tw, d.camera.to_world, sizeof k.cam.tw);
    k.cam.near_clip = d.camera.near_clip; k.cam.far_clip = d.camera.far_clip;
    k.inv_w = 1.0f / (float)d.film.width; k.inv_h = 1.0f / (float)d.film.height;
    k.sensor_confocal = n->sensor_is_confocal ? 1u : 0u;
    k.sensor_target = ld3(n->sensor_target);
    if (n->sensor_is_confocal && (camera_sensor || d.film.width != 1 || d.film.height != 1))
        return "Confocal configuration requires a nlos_capture_meter with a film of size [1,1]";
    if (!camera_sensor) {
        const mtr_shape &rw = n->shapes[n->relay_shape];
        k.w_center = ld3(rw.center); k.w_du = ld3(rw.du); k.w_dv = ld3(rw.dv);
    } else { k.w_center = mk(0, 0, 0); k.w_du = mk(1, 0, 0); k.w_dv = mk(0, 1, 0); }
    const float *T = n->laser_to_world;
    k.l_origin = mk(T[3], T[7], T[11]);
    k.l_forward = mk(T[2], T[6], T[10]);
    const float inv[9] = { T[0], T[4], T[8], T[1], T[5], T[9], T[2], T[6], T[10] };      // rigid: inverse rotation = transpose
    memcpy(k.l_inv, inv, sizeof inv);
    k.l_cot = (float)(1.0 / tan(0.5 * (double)n->laser_fov * 3.14159265358979323846 / 180.0));
    k.l_scale = n->laser_scale;
    k.l_irr = ld3(n->laser_irradiance);
    k.capture_type = n->capture_type; k.flags = n->flags; k.filter_depth = n->filter_depth; k.n_shapes = n->n_shapes;
    k.film_w = d.film.width; k.film_h = d.film.height;
    k.laser_w = exhaustive ? d.film.laser_scan_width : 0u; k.laser_h = exhaustive ? d.film.laser_scan_height : 0u;
    k.illum_tan = (float)tan(0.5 * (double)n->illumination_scan_fov * 3.14159265358979323846 / 180.0);
    const float rot[9] = { T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10] };
    memcpy(k.l_rot, rot, sizeof rot);

    const uint32_t ns = n->n_shapes;
    o.shapes.assign(ns, NlosShape{});
    o.shape_pmf.assign(ns, 0.0f); o.shape_cdf.assign(ns, 0.0f);
    o.face_pmf.assign(d.n_tris ? d.n_tris : 1, 0.0f); o.face_cdf.assign(d.n_tris ? d.n_tris : 1, 0.0f);
    o.hg_tris.assign(3 * (size_t)(d.n_tris ? d.n_tris : 1), q4{ 0, 0, 0, 0 });
    std::vector<double> area(ns, 0.0);
    double total = 0.0;
    for (uint32_t s = 0; s < ns; ++s) {
        const mtr_shape &S = n->shapes[s];
        NlosShape &D = o.shapes[s];
        D.first_tri = S.first_tri; D.n_tris = S.n_tris; D.is_rect = S.is_rectangle;
        double a = fill_mesh_tables(d.tri_verts, S.first_tri, S.n_tris, o.face_pmf.data(), o.face_cdf.data(), o.hg_tris.data());
        if (!S.is_rectangle) fill_mesh_normals(d.tri_normals, S.first_tri, S.n_tris, d.n_tris, o.hg_vn);
        if (S.is_rectangle) {
            for (int c = 0; c < 3; ++c) { D.center[c] = S.center[c]; D.du[c] = S.du[c]; D.dv[c] = S.dv[c]; }
            const f3 cr = cross(ld3(S.du), ld3(S.dv));
            const double len = sqrt((double)cr.x * cr.x + (double)cr.y * cr.y + (double)cr.z * cr.z);
            a = 4.0 * len;
            f3 nn = cr / sqrtf(dot(cr, cr));
            if (S.is_rectangle & MTR_RECT_FLIP_NORMALS) nn = mk(-nn.x, -nn.y, -nn.z);
            D.n[0] = nn.x; D.n[1] = nn.y; D.n[2] = nn.z;
        }
        D.inv_area = (float)(1.0 / a);
        // transientnlospath.py:277-292: the relay wall has weight 0 unless ..._includes_relay_wall
        area[s] = (s == n->relay_shape && !(n->flags & MTR_NLOS_HG_INCLUDES_WALL)) ? 0.0 : a;
        total += area[s];
    }
    if ((n->flags & MTR_NLOS_HG_SAMPLING) && !(total > 0.0))
        return "Hidden geometry sampling is activated, but the hidden geometry in the scene has zero surface area?";
    double acc = 0.0;
    for (uint32_t s = 0; s < ns; ++s) {
        acc += area[s];
        o.shape_pmf[s] = total > 0.0 ? (float)(area[s] / total) : 0.0f;
        o.shape_cdf[s] = total > 0.0 ? (float)(acc / total) : 0.0f;
    }
    return nullptr;
}

RenderConst make_render_const(const mtr_render_params &p, const Film &f, uint32_t n_emitters)
{
    RenderConst rc{};
    rc.div_crop_w = fastdiv_make(f.crop_w);
    rc.spp_total = p.spp_total; rc.seed = p.seed;
    rc.max_depth = p.max_depth < 0 ? 0xffffffffu : (uint32_t)p.max_depth;
    rc.rr_depth = (uint32_t)p.rr_depth; rc.flags = p.flags;
    rc.sample_scale = (float)(1.0 / (double)(p.spp_scale ? p.spp_scale : p.spp_total));          // common.py:173-175 (total_spp of all passes)
    rc.inv_crop_w = 1.0f / (float)f.crop_w; rc.inv_crop_h = 1.0f / (float)f.crop_h;
    rc.off_x = -(float)f.crop_x * rc.inv_crop_w; rc.off_y = -(float)f.crop_y * rc.inv_crop_h;
    rc.n_emitters_f = (float)n_emitters;
    rc.inv_n_emitters = n_emitters ? 1.0f / (float)n_emitters : 0.0f;
    return rc;
}

} // namespace mtr
