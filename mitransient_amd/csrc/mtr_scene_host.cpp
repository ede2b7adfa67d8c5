#include "mtr_scene_host.h"
#include <cstring>
#include <cmath>

namespace mtr {

Film film_from_desc(const mtr_film_desc &d)
{
    Film f;
    f.width = d.width; f.height = d.height; f.crop_w = d.crop_width; f.crop_h = d.crop_height;
    f.crop_x = d.crop_offset_x; f.crop_y = d.crop_offset_y; f.bins = d.temporal_bins;
    f.start_opl = d.start_opl; f.bin_width = d.bin_width_opl;
    return f;
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
        if (d.materials[i].type > MTR_BSDF_NULL) return "unknown BSDF type";

    s.film = film_from_desc(d.film);
    memcpy(s.cam.s2c, d.camera.sample_to_camera, sizeof s.cam.s2c);
    memcpy(s.cam.tw, d.camera.to_world, sizeof s.cam.tw);
    s.cam.near_clip = d.camera.near_clip; s.cam.far_clip = d.camera.far_clip;

    // BVH2 over the triangles; triangles are stored in leaf order
    BvhBuild bvh;
    build_bvh(d.tri_verts, d.n_tris, bvh);
    s.nodes = bvh.nodes; s.bvh_depth = bvh.max_depth; s.n_leaves = bvh.n_leaves;
    s.tgeom.resize(d.n_tris); s.tshade.resize(d.n_tris);
    for (uint32_t slot = 0; slot < d.n_tris; ++slot) {
        const uint32_t o = bvh.order[slot];
        const float *v = d.tri_verts + 9 * (size_t)o;
        TriGeom &g = s.tgeom[slot]; TriShade &h = s.tshade[slot];
        const uint32_t mat_em = d.tri_material[o] | ((uint32_t)(d.tri_emitter[o] + 1) << 16);
        // flat frame: n = normalize(e1 x e2), s = normalize(e1), t = n x s   (f32, contract in DESIGN.md)
        f3 p0 = mk(v[0], v[1], v[2]), e1 = mk(v[3], v[4], v[5]) - p0, e2 = mk(v[6], v[7], v[8]) - p0;
        f3 n = normalize(cross(e1, e2)), sdir = normalize(e1), t = cross(n, sdir);
        g.g[0] = q4{ p0.x, p0.y, p0.z, e1.x };
        g.g[1] = q4{ e1.y, e1.z, e2.x, e2.y };
        g.g[2] = q4{ e2.z, bitsf(o), bitsf(mat_em), 0.0f };
        h.h[0] = q4{ n.x, n.y, n.z, sdir.x };
        h.h[1] = q4{ sdir.y, sdir.z, t.x, t.y };
        h.h[2] = q4{ t.z, v[3], v[4], v[5] };
        h.h[3] = q4{ v[6], v[7], v[8], 0.0f };
    }
    s.ems.resize(d.n_emitters);
    for (uint32_t i = 0; i < d.n_emitters; ++i) {
        const mtr_emitter &e = d.emitters[i];
        Emitter &E = s.ems[i];
        for (int k = 0; k < 3; ++k) { E.center[k] = e.center[k]; E.du[k] = e.du[k]; E.dv[k] = e.dv[k]; E.radiance[k] = e.radiance[k]; }
        f3 cr = cross(ld3(e.du), ld3(e.dv));
        float len = sqrtf(dot(cr, cr));
        f3 n = cr / len;
        E.n[0] = n.x; E.n[1] = n.y; E.n[2] = n.z;
        E.inv_area = 1.0f / (4.0f * len);      // rectangle area = |(2 du) x (2 dv)|
    }
    s.mats.assign(d.materials, d.materials + d.n_materials);
    return nullptr;
}

RenderConst make_render_const(const mtr_render_params &p, const Film &f, uint32_t n_emitters)
{
    RenderConst rc{};
    rc.spp_total = p.spp_total; rc.seed = p.seed;
    rc.max_depth = p.max_depth < 0 ? 0xffffffffu : (uint32_t)p.max_depth;
    rc.rr_depth = (uint32_t)p.rr_depth; rc.flags = p.flags;
    rc.sample_scale = (float)(1.0 / (double)p.spp_total);          // common.py:173-175
    rc.inv_crop_w = 1.0f / (float)f.crop_w; rc.inv_crop_h = 1.0f / (float)f.crop_h;
    rc.off_x = -(float)f.crop_x * rc.inv_crop_w; rc.off_y = -(float)f.crop_y * rc.inv_crop_h;
    rc.n_emitters_f = (float)n_emitters;
    rc.inv_n_emitters = n_emitters ? 1.0f / (float)n_emitters : 0.0f;
    return rc;
}

} // namespace mtr
