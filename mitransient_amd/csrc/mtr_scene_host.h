// mtr_scene_host.h — host-side scene ingestion: mtr_scene_desc -> device-layout arrays
// (BVH2 node packets, leaf-ordered triangles with flat shading frames, emitters with derived
// normal / inverse area).  Pure C++, no HIP: the API layer uploads the result; the test host
// harness runs the same arrays through mtr_core.h on the CPU.
#pragma once
#include <vector>
#include "mtr_core.h"
#include "mtr_bvh.h"
#include "mtr_nlos.h"

namespace mtr {

struct HostScene {
    std::vector<Node> nodes;
    std::vector<WNode> wnodes;                 // 8-wide collapse of `nodes` (small scenes only: walked in LDS)
    std::vector<QNode8> wnodes8q;              // quantised 8-wide collapse (large scenes only)
    std::vector<QNode4> wnodes4;               // quantised 4-wide collapse of `nodes` (walked in HBM by the wavefront kernels)
    bool has_wide = false;                     // wnodes is valid (possibly empty: a scene without triangles)
    std::vector<TriPair> tpairs;               // [n_slots / 2]
    std::vector<TriShade> tshade;              // [n_slots]
    std::vector<uint32_t> slot_orig;           // [n_slots] original triangle index (pad slots: the triangle they repeat)
    std::vector<mtr_material> mats;
    std::vector<Emitter> ems;
    std::vector<q4> samp_tris;                 // mesh emitters only (empty otherwise): see SceneView
    std::vector<q4> samp_vn;                   // ... their vertex normals (empty unless a mesh emitter has them): mesh_sample_position
    std::vector<q4> vnormals;                  // [3 * n_slots] vertex normals, when a triangle is smooth-shaded (empty otherwise)
    // bitmap textures (empty without): texels as RGBA f32 of all textures, (first texel, width, height, 0) per texture, and
    // the corner texture coordinates by slot, two quads each
    std::vector<q4> texels; std::vector<q4> tex_info; std::vector<q4> uvs;
    std::vector<float> face_pmf, face_cdf;
    uint32_t bvh_depth = 0, n_leaves = 0;
    uint32_t wide_levels = 0, wide4_levels = 0, wide8q_levels = 0;        // levels of the collapsed trees (= their traversal stack bound)
    Camera cam{};
    Film film{};
};

// NLOS tier tables (TransientNLOSPath.prepare, transientnlospath.py:251-292): shape / face distributions,
// rectangle normals, triangles in ORIGINAL order for Mesh::sample_position, projector constants
struct HostNlos {
    std::vector<NlosShape> shapes;
    std::vector<float> shape_pmf, shape_cdf, face_pmf, face_cdf;
    std::vector<q4> hg_tris;
    std::vector<q4> hg_vn;                     // vertex normals of hidden meshes that have them (empty otherwise): mesh_sample_position
    NlosConst k{};            // pointers left null: the caller points them at its copies
};
const char *derive_nlos(const mtr_scene_desc &d, HostNlos &out);

Film film_from_desc(const mtr_film_desc &d);
// returns nullptr on success or a static error string
const char *derive_scene(const mtr_scene_desc &d, HostScene &out);
RenderConst make_render_const(const mtr_render_params &p, const Film &f, uint32_t n_emitters);

} // namespace mtr
