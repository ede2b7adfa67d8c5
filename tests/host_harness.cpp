// host_harness.cpp — TEST-ONLY.  Compiles the product's per-path arithmetic (mtr_core.h, the
// BVH builder and scene ingestion) for the HOST and runs it one lane at a time, so that the
// CPU test-suite can compare it against the oracle before any GPU time is spent.  It is never
// built into libmitransient_amd.so and nothing in mitransient_amd/ can reach it: the product
// has no CPU path.
#include "../mitransient_amd/csrc/mtr_core.h"
#include "../mitransient_amd/csrc/mtr_scene_host.h"
#include "../mitransient_amd/csrc/mtr_nlos.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

using namespace mtr;

#ifdef HH_WALK_STATS          // tools/walk_stats.py: node steps / primitive passes of the host walk (one thread)
static uint64_t g_walk_steps[2];
extern "C" void hh_walk_steps(uint64_t *out, int reset) { out[0] = g_walk_steps[0]; out[1] = g_walk_steps[1]; if (reset) g_walk_steps[0] = g_walk_steps[1] = 0; }
#endif
namespace {
struct ArrStack {
    static constexpr bool kPark = false;      // (k_fused's stack can park path state in LDS: mtr_kernels.hip)
    void park_prev_p(mtr::f3) {}
    mtr::f3 unpark_prev_p() const { return mtr::mk(0, 0, 0); }
    void park_inc(uint64_t) {}
    uint64_t unpark_inc() const { return 0; }
    void park_prev_pdf(float) {}
    float unpark_prev_pdf() const { return 0.0f; }
    int32_t v[130]; int sp;
    void reset() { sp = 0; }
    void push_if(bool c, int32_t x) { v[sp] = x; sp += c ? 1 : 0; }
    int32_t pop() { return v[--sp]; }
    bool empty() const { return sp == 0; }
    void prof_mark(int) {}
#ifdef HH_WALK_STATS
    void count(int k) { ++g_walk_steps[k]; }
#else
    void count(int) {}
#endif
};
struct HostSink {
    float *film; uint32_t W, T; uint64_t n;
    Film fm;
    void splat(uint32_t fx, uint32_t fy, uint32_t bin, float r, float g, float b, float opl, uint32_t, uint32_t)
    {
        if (fm.n_freq) {                        // phasor film: the arithmetic of k_wf_phasor_scatter, contribution by contribution
            float *dst = film + ((size_t)fy * W + fx) * (2u * fm.n_freq + 1u);
            const float rel = opl - fm.start_opl;
            for (uint32_t f = 0; f < fm.n_freq; ++f) {
                float c, s;
                phasor_term(fm.freq[f], rel, c, s);
                dst[2 * f] += r * c; dst[2 * f + 1] += r * s;
            }
            ++n;
            return;
        }
        size_t idx = (((size_t)fy * W + fx) * T + bin) * 4u;
        film[idx] += r; film[idx + 1] += g; film[idx + 2] += b; ++n;
    }
};
}

// both plane-fetch forms of trav_node_step (selects / sign-dependent offsets) are run through the CPU tests
static bool g_node_pairs = false;
extern "C" void hh_set_node_pairs(int on) { g_node_pairs = on != 0; }
// ... and so is the 8-wide tree the fused kernel walks in LDS
static int g_wide = 0;       // 1: 8-wide (LDS form), 2: quantised 4-wide, 3: quantised 8-wide (HBM forms)
extern "C" void hh_set_wide(int on) { g_wide = on; }

extern "C" int hh_render(const mtr_scene_desc *d, const mtr_render_params *p, float *t4, float *s4, mtr_counters *out)
{
    HostScene hs;
    if (derive_scene(*d, hs)) return -1;
    SceneView sv;
    sv.nodes = hs.nodes.data(); sv.tpairs = hs.tpairs.data(); sv.tshade = hs.tshade.data();
    sv.node_pairs = g_node_pairs;
    sv.wnodes = (g_wide == 1 && hs.has_wide && !hs.wnodes.empty()) ? hs.wnodes.data() : nullptr;
    sv.wnodes4 = (g_wide == 2 && !hs.wnodes4.empty()) ? hs.wnodes4.data() : nullptr;
    sv.wnodes8q = (g_wide == 3 && !hs.wnodes8q.empty()) ? hs.wnodes8q.data() : nullptr;
    sv.mats = hs.mats.data(); sv.ems = hs.ems.data();
    sv.n_emitters = (uint32_t)hs.ems.size(); sv.n_slots = (uint32_t)hs.tshade.size();
    sv.samp_tris = hs.samp_tris.data(); sv.samp_vn = hs.samp_vn.empty() ? nullptr : hs.samp_vn.data(); sv.face_pmf = hs.face_pmf.data(); sv.face_cdf = hs.face_cdf.data();
    sv.vnormals = hs.vnormals.empty() ? nullptr : hs.vnormals.data();
    sv.texels = hs.texels.empty() ? nullptr : hs.texels.data(); sv.tex_info = hs.tex_info.empty() ? nullptr : hs.tex_info.data();
    sv.uvs = hs.uvs.empty() ? nullptr : hs.uvs.data();
    RenderConst rc = make_render_const(*p, hs.film, sv.n_emitters);
    HostSink sink{ t4, hs.film.width, hs.film.bins, 0, hs.film };
    ArrStack st; st.sp = 0;
    uint64_t closest = 0, shadow = 0, bounces = 0, paths = 0;
    // NLOS tier: tables + scanned points (the product computes the latter in k_nlos_prepare)
    HostNlos hn; std::vector<q4> targets;
    const bool nlos = d->nlos != nullptr;
    // the product's rule for the extended shading code (mtr_api.hip: has_rough): a GGX lobe, a smooth-shaded triangle, a bitmap
    bool ext = !hs.vnormals.empty() || !hs.texels.empty();
    for (uint32_t i = 0; i < d->n_materials; ++i) ext = ext || bsdf_is_rough(d->materials[i].type);
    if (nlos) {
        if (derive_nlos(*d, hn)) return -2;
        NlosConst &k = hn.k;
        k.shapes = hn.shapes.data(); k.shape_pmf = hn.shape_pmf.data(); k.shape_cdf = hn.shape_cdf.data();
        k.face_pmf = hn.face_pmf.data(); k.face_cdf = hn.face_cdf.data(); k.hg_tris = hn.hg_tris.data(); k.hg_vn = hn.hg_vn.empty() ? nullptr : hn.hg_vn.data();
        const uint32_t n = nlos_target_count(k);
        targets.resize(n);
        for (uint32_t i = 0; i < n; ++i) {
            const Ray r = nlos_prepare_ray(k, i);
            Hit h = traverse<false>(sv, r.o, r.d, r.tmax, st);
            f3 pp = mk(0, 0, 0);
            if (h.prim >= 0) pp = hit_ctx<false>(sv, r.d, h).sp;
            targets[i] = q4{ pp.x, pp.y, pp.z, 0.0f };
        }
        k.targets = targets.data();
    }
    for (uint32_t pix = p->pixel_begin; pix < p->pixel_end; ++pix)
        for (uint32_t s = p->spp_begin; s < p->spp_end; ++s) {
            Path path;
            if (nlos) {
                nlos_begin(path, hn.k, hs.film, rc, pix, s);
                ++paths;
                bool alive = true;
                while (alive) {
                    BounceStats bs{ 0, 0 };
                    alive = ext ? nlos_bounce<true>(path, sv, hn.k, hs.film, rc, st, sink, bs) : nlos_bounce<false>(path, sv, hn.k, hs.film, rc, st, sink, bs);
                    closest += bs.closest; shadow += bs.shadow; ++bounces;
                }
                uint32_t fx = path.px - hs.film.crop_x, fy = path.py - hs.film.crop_y;
                if (fx < hs.film.width && fy < hs.film.height) {
                    float *sp = s4 + ((size_t)fy * hs.film.width + fx) * 4u;
                    sp[0] += path.L.x; sp[1] += path.L.y; sp[2] += path.L.z; sp[3] += 1.0f;
                }
                continue;
            }
            path_begin(path, hs.cam, hs.film, rc, pix, s);
            ++paths;
            if (rc.flags & MTR_FLAG_CAMERA_UNWARP) {
                Hit h0 = traverse<false>(sv, path.ray.o, path.ray.d, path.ray.tmax, st);
                ++closest;
                if (h0.prim >= 0) path.dist = -h0.t;
            }
            bool alive = true;
            const bool trace_log = getenv("HH_TRACE_LOG") != nullptr;       // debugging aid: the ray of every bounce
            while (alive) {
                BounceStats bs{ 0, 0 };
                if (trace_log) printf("s %u depth %u ray o %.9g %.9g %.9g d %.9g %.9g %.9g tmax %.9g\n", s, path.depth, path.ray.o.x, path.ray.o.y, path.ray.o.z, path.ray.d.x, path.ray.d.y, path.ray.d.z, path.ray.tmax);
                alive = path_bounce(path, sv, hs.film, rc, st, sink, bs);
                closest += bs.closest; shadow += bs.shadow; ++bounces;
            }
            uint32_t fx = path.px - hs.film.crop_x, fy = path.py - hs.film.crop_y;
            if (fx < hs.film.width && fy < hs.film.height) {
                float *sp = s4 + ((size_t)fy * hs.film.width + fx) * 4u;
                sp[0] += path.L.x; sp[1] += path.L.y; sp[2] += path.L.z; sp[3] += 1.0f;
            }
        }
    if (out) {
        memset(out, 0, sizeof *out);
        out->paths = paths; out->rays_closest = closest; out->rays_shadow = shadow;
        out->bounces = bounces; out->splats_issued = sink.n;
    }
    return 0;
}

extern "C" int hh_bvh_info(const mtr_scene_desc *d, uint32_t *n_nodes, uint32_t *depth, uint32_t *leaves)
{
    HostScene hs;
    if (derive_scene(*d, hs)) return -1;
    *n_nodes = (uint32_t)hs.nodes.size(); *depth = hs.bvh_depth; *leaves = hs.n_leaves;
    return 0;
}

// leaves of the quantised 8-wide tree by triangle count (1 .. 4), rectangles in [0]; nodes and children in [5], [6]
extern "C" int hh_leaf_sizes(const mtr_scene_desc *d, uint64_t *out7)
{
    HostScene hs;
    if (derive_scene(*d, hs)) return -1;
    for (int k = 0; k < 7; ++k) out7[k] = 0;
    for (const QNode8 &n : hs.wnodes8q) {
        const uint32_t cnt = (fbits(n.q[0].w) >> 26) & 0xfu;
        ++out7[5]; out7[6] += cnt;
        for (uint32_t c = 0; c < cnt; ++c) {
            const int32_t ref = *((const int32_t *)&n.q[4] + c);
            if (ref >= 0) continue;
            const uint32_t code = ~(uint32_t)ref;
            ++out7[(code & kLeafQuadBit) ? 0u : (code & 3u) + 1u];
        }
    }
    return 0;
}

// FNV-1a over everything the builders produce (BVH2 packets, the quantised trees, the slot order): two builds of one scene
// with different thread counts must agree on it
extern "C" int hh_tree_hash(const mtr_scene_desc *d, uint64_t *hash)
{
    HostScene hs;
    if (derive_scene(*d, hs)) return -1;
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](const void *p, size_t n) { const unsigned char *b = (const unsigned char *)p; for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; } };
    mix(hs.nodes.data(), hs.nodes.size() * sizeof(Node));
    mix(hs.wnodes4.data(), hs.wnodes4.size() * sizeof(QNode4));
    mix(hs.wnodes8q.data(), hs.wnodes8q.size() * sizeof(QNode8));
    mix(hs.slot_orig.data(), hs.slot_orig.size() * sizeof(uint32_t));
    mix(hs.tpairs.data(), hs.tpairs.size() * sizeof(TriPair));
    *hash = h;
    return 0;
}

// closest hit / occlusion through the product's BVH, for BVH-vs-brute-force tests
extern "C" int hh_intersect(const mtr_scene_desc *d, uint32_t n, const float *o3, const float *d3, const float *maxt,
                            float *t_out, int32_t *prim_out, uint8_t *occ_out)
{
    HostScene hs;
    if (derive_scene(*d, hs)) return -1;
    SceneView sv;
    sv.nodes = hs.nodes.data(); sv.tpairs = hs.tpairs.data(); sv.tshade = hs.tshade.data();
    sv.node_pairs = g_node_pairs;
    sv.wnodes = (g_wide == 1 && hs.has_wide && !hs.wnodes.empty()) ? hs.wnodes.data() : nullptr;
    sv.wnodes4 = (g_wide == 2 && !hs.wnodes4.empty()) ? hs.wnodes4.data() : nullptr;
    sv.wnodes8q = (g_wide == 3 && !hs.wnodes8q.empty()) ? hs.wnodes8q.data() : nullptr;
    sv.mats = hs.mats.data(); sv.ems = hs.ems.data();
    sv.n_emitters = (uint32_t)hs.ems.size(); sv.n_slots = (uint32_t)hs.tshade.size();
    sv.samp_tris = hs.samp_tris.data(); sv.samp_vn = hs.samp_vn.empty() ? nullptr : hs.samp_vn.data(); sv.face_pmf = hs.face_pmf.data(); sv.face_cdf = hs.face_cdf.data();
    sv.vnormals = hs.vnormals.empty() ? nullptr : hs.vnormals.data();
    sv.texels = hs.texels.empty() ? nullptr : hs.texels.data(); sv.tex_info = hs.tex_info.empty() ? nullptr : hs.tex_info.data();
    sv.uvs = hs.uvs.empty() ? nullptr : hs.uvs.data();
    ArrStack st; st.sp = 0;
    for (uint32_t i = 0; i < n; ++i) {
        f3 o = mk(o3[3 * i], o3[3 * i + 1], o3[3 * i + 2]), dd = mk(d3[3 * i], d3[3 * i + 1], d3[3 * i + 2]);
        float mt = maxt ? maxt[i] : kInf;
        Hit h = traverse<false>(sv, o, dd, mt, st);
        t_out[i] = h.t; prim_out[i] = h.prim >= 0 ? (int32_t)hs.slot_orig[h.prim] : -1;
        Hit a = traverse<true>(sv, o, dd, mt, st);
        occ_out[i] = a.prim >= 0;
    }
    return 0;
}

// the product's restatements of exp / log / erf / erfinv (Beckmann lobes): which = 0 exp, 1 log, 2 erf, 3 erfinv
extern "C" void hh_special(int which, uint64_t n, const float *x, float *y)
{
    for (uint64_t i = 0; i < n; ++i)
        y[i] = which == 0 ? mtr_expf(x[i]) : which == 1 ? mtr_logf(x[i]) : which == 2 ? mtr_erff(x[i]) : mtr_erfinvf(x[i]);
}

// the product's BSDF arithmetic on arrays of local directions (tests/test_rough_bsdf.py)
extern "C" void hh_bsdf_eval_pdf(const mtr_material *m, uint32_t n, const float *wi3, const float *wo3, float *val3, float *pdf)
{
    for (uint32_t i = 0; i < n; ++i) {
        f3 wi = mk(wi3[3 * i], wi3[3 * i + 1], wi3[3 * i + 2]), wo = mk(wo3[3 * i], wo3[3 * i + 1], wo3[3 * i + 2]);
        f3 v = mk(0, 0, 0); float p = 0.0f;
        if ((m->flags & MTR_MAT_TWOSIDED) && wi.z < 0.0f) { wi.z = -wi.z; wo.z = -wo.z; }
        if (bsdf_is_rough(m->type)) rough_eval_pdf(*m, mk(m->a[0], m->a[1], m->a[2]), wi, wo, v, p);
        else if (m->type == MTR_BSDF_DIFFUSE && wi.z > 0.0f && wo.z > 0.0f) {
            p = kInvPi * wo.z; v = mk((m->a[0] * kInvPi) * wo.z, (m->a[1] * kInvPi) * wo.z, (m->a[2] * kInvPi) * wo.z);
        }
        val3[3 * i] = v.x; val3[3 * i + 1] = v.y; val3[3 * i + 2] = v.z; pdf[i] = p;
    }
}
extern "C" void hh_bsdf_sample(const mtr_material *m, uint32_t n, const float *wi3, const float *u1, const float *ua, const float *ub,
                               float *wo3, float *pdf, float *w3)
{
    for (uint32_t i = 0; i < n; ++i) {
        const BsdfSample bs = bsdf_sample<true>(*m, mk(wi3[3 * i], wi3[3 * i + 1], wi3[3 * i + 2]), u1[i], ua[i], ub[i], mk(m->a[0], m->a[1], m->a[2]));
        wo3[3 * i] = bs.wo.x; wo3[3 * i + 1] = bs.wo.y; wo3[3 * i + 2] = bs.wo.z; pdf[i] = bs.pdf;
        w3[3 * i] = bs.w.x; w3[3 * i + 1] = bs.w.y; w3[3 * i + 2] = bs.w.z;
    }
}

// debugging aid: prints the 8-wide tree of a scene (child counts, leaf / inner refs, walk axis)
extern "C" int hh_print_wide(const mtr_scene_desc *d)
{
    HostScene hs;
    if (derive_scene(*d, hs)) return -1;
    printf("bvh2 packets %zu, wide nodes %zu, slots %zu\n", hs.nodes.size(), hs.wnodes.size(), hs.tshade.size());
    for (size_t i = 0; i < hs.wnodes.size(); ++i) {
        const WNode &w = hs.wnodes[i];
        printf("  wnode %zu axis %u count %u quads %u%s:", i, w.axis, w.count, w.n_quads, (w.flags & 2u) ? " BOX" : ((w.flags & 1u) ? " OBJECT" : ""));
        const float *f = &w.box[0].x;
        for (uint32_t c = 0; c < w.count; ++c) {
            if (w.ref[c] >= 0) printf(" N%d", w.ref[c]);
            else { uint32_t code = ~(uint32_t)w.ref[c]; printf(" %c%u(%u)", (code & kLeafQuadBit) ? 'Q' : 'L', (code & ~kLeafQuadBit) >> 2, (code & 3u) + 1u); }
            const int j = (int)c >> 1, hh = (int)c & 1;
            printf("[%.2f %.2f %.2f]", f[4 * (3 * j + 0) + 2 + hh] - f[4 * (3 * j + 0) + hh], f[4 * (3 * j + 1) + 2 + hh] - f[4 * (3 * j + 1) + hh],
                   f[4 * (3 * j + 2) + 2 + hh] - f[4 * (3 * j + 2) + hh]);
        }
        printf("\n");
    }
    return 0;
}

// number of BOX nodes of the 8-wide tree (mtr_core.h box_select); -2 when one of them is not six two-triangle leaves
extern "C" int hh_count_box_nodes(const mtr_scene_desc *d)
{
    HostScene hs;
    if (derive_scene(*d, hs)) return -1;
    int n = 0;
    for (const WNode &w : hs.wnodes) {
        if (!(w.flags & 2u)) continue;
        if (!(w.flags & 1u) || w.count != 6u || w.n_quads != 0u) return -2;
        for (uint32_t c = 0; c < 6u; ++c) if (w.ref[c] >= 0 || ((~(uint32_t)w.ref[c]) & 3u) != 1u || ((~(uint32_t)w.ref[c]) & kLeafQuadBit)) return -2;
        ++n;
    }
    return n;
}

// structural checks of the collapsed trees (tests/test_scene_host.py): every BVH2 leaf is referenced exactly once by the
// 8-wide and by the quantised 4-wide tree, every wide node is reachable exactly once, and each quantised child box
// contains the (padded) BVH2 box it was made from.  Returns 0 or a negative code naming the first violation.
namespace {
void bvh2_leaves(const std::vector<Node> &nodes, std::vector<int32_t> &out)
{
    if (nodes.empty()) return;
    std::vector<int32_t> st{ 0 };
    while (!st.empty()) {
        const int32_t n = st.back(); st.pop_back();
        const float *f = &nodes[n].q[0].x;
        for (int c = 0; c < 2; ++c) {
            if (!(f[c] <= f[2 + c])) continue;                       // absent child
            const int32_t ref = (int32_t)fbits(f[12 + c]);
            if (ref >= 0) st.push_back(ref); else out.push_back(ref);
        }
    }
}
}
extern "C" int hh_check_wide(const mtr_scene_desc *d, uint32_t *n_wide8, uint32_t *n_wide4)
{
    HostScene hs;
    if (derive_scene(*d, hs)) return -1;
    std::vector<int32_t> want;
    bvh2_leaves(hs.nodes, want);
    std::sort(want.begin(), want.end());
    if (n_wide8) *n_wide8 = (uint32_t)hs.wnodes.size();
    if (n_wide4) *n_wide4 = (uint32_t)hs.wnodes4.size();
    if (hs.has_wide) {
        std::vector<int32_t> got; std::vector<int> seen(hs.wnodes.size(), 0);
        std::vector<int32_t> st; if (!hs.wnodes.empty()) { st.push_back(0); seen[0] = 1; }
        while (!st.empty()) {
            const WNode &w = hs.wnodes[st.back()]; st.pop_back();
            if (w.count < 1 || w.count > kWide || w.axis > 2) return -2;
            for (uint32_t c = 0; c < w.count; ++c) {
                const bool quad = w.ref[c] < 0 && ((~(uint32_t)w.ref[c]) & kLeafQuadBit) != 0u;
                if (quad != (c < w.n_quads)) return -3;                  // rectangle children come first, and only they
                if ((w.flags & 1u) && w.ref[c] >= 0) return -7;          // an object node holds leaves only
                if (w.ref[c] < 0) got.push_back(w.ref[c]);
                else { if ((size_t)w.ref[c] >= seen.size() || seen[w.ref[c]]++) return -4; st.push_back(w.ref[c]); }
            }
        }
        std::sort(got.begin(), got.end());
        if (got != want) return -5;
        for (int v : seen) if (v != 1) return -6;
    }
    {
        std::vector<int32_t> got; std::vector<int> seen(hs.wnodes4.size(), 0);
        struct Item { int32_t node; };
        std::vector<int32_t> st; if (!hs.wnodes4.empty()) { st.push_back(0); seen[0] = 1; }
        while (!st.empty()) {
            const QNode4 &q = hs.wnodes4[st.back()]; st.pop_back();
            const uint32_t meta = fbits(q.q[0].w), count = meta >> 26;
            if (count < 1 || count > 4 || ((meta >> 24) & 3u) > 2) return -12;
            for (uint32_t c = 0; c < count; ++c) {
                const int32_t ref = (int32_t)fbits((&q.q[1].x)[c]);
                if (ref < 0) got.push_back(ref);
                else { if ((size_t)ref >= seen.size() || seen[ref]++) return -14; st.push_back(ref); }
            }
        }
        std::sort(got.begin(), got.end());
        if (got != want) return -15;
        for (int v : seen) if (v != 1) return -16;
    }
    {   // the quantised 8-wide tree: every leaf exactly once, every node reached once
        std::vector<int32_t> got; std::vector<int> seen(hs.wnodes8q.size(), 0);
        std::vector<int32_t> st; if (!hs.wnodes8q.empty()) { st.push_back(0); seen[0] = 1; }
        while (!st.empty()) {
            const QNode8 &q = hs.wnodes8q[st.back()]; st.pop_back();
            const uint32_t meta = fbits(q.q[0].w), count = (meta >> 26) & 0xfu;
            if (count < 1 || count > 8 || ((meta >> 24) & 3u) > 2) return -32;
            for (uint32_t c = 0; c < count; ++c) {
                const int32_t ref = (int32_t)fbits((&q.q[4].x)[c]);
                if (ref < 0) got.push_back(ref);
                else { if ((size_t)ref >= seen.size() || seen[ref]++) return -34; st.push_back(ref); }
            }
        }
        std::sort(got.begin(), got.end());
        if (!hs.wnodes8q.empty() && got != want) return -35;
        for (int v : seen) if (v != 1) return -36;
    }
    // containment: walk BVH2 and the 4-wide tree together is not possible (different shapes); instead check, for every
    // 4-wide node, that each decoded child box contains the union of the leaf boxes below it — cheap proxy: it must contain
    // the decoded boxes of that child's own children (inner) — and that the root's children cover every triangle vertex.
    if (!hs.wnodes4.empty()) {
        auto decode = [&](const QNode4 &q, uint32_t c, double lo[3], double hi[3]) {
            const uint32_t meta = fbits(q.q[0].w);
            const float org[3] = { q.q[0].x, q.q[0].y, q.q[0].z };
            const uint32_t wl[3] = { fbits(q.q[2].x), fbits(q.q[2].y), fbits(q.q[2].z) }, wh[3] = { fbits(q.q[2].w), fbits(q.q[3].x), fbits(q.q[3].y) };
            for (int k = 0; k < 3; ++k) {
                const double step = std::ldexp(1.0, (int)((meta >> (8 * k)) & 0xffu) - 127);
                lo[k] = (double)org[k] + (double)((wl[k] >> (8 * c)) & 0xffu) * step;
                hi[k] = (double)org[k] + (double)((wh[k] >> (8 * c)) & 0xffu) * step;
            }
        };
        for (size_t i = 0; i < hs.wnodes4.size(); ++i) {
            const QNode4 &q = hs.wnodes4[i];
            const uint32_t count = fbits(q.q[0].w) >> 26;
            for (uint32_t c = 0; c < count; ++c) {
                double lo[3], hi[3]; decode(q, c, lo, hi);
                const int32_t ref = (int32_t)fbits((&q.q[1].x)[c]);
                if (ref >= 0) {
                    const QNode4 &ch = hs.wnodes4[ref];
                    const uint32_t cc = fbits(ch.q[0].w) >> 26;
                    for (uint32_t k = 0; k < cc; ++k) {
                        double l2[3], h2[3]; decode(ch, k, l2, h2);
                        // planes round outwards on the child's own grid: allow one step of that grid (plus the box padding) as slack
                        const uint32_t cm = fbits(ch.q[0].w);
                        for (int a = 0; a < 3; ++a) {
                            const double slack = std::ldexp(1.0, (int)((cm >> (8 * a)) & 0xffu) - 127) + 1e-4 * (1.0 + std::fabs(hi[a]));
                            if (l2[a] < lo[a] - slack || h2[a] > hi[a] + slack) return -20;
                        }
                    }
                } else {
                    const uint32_t code = (~(uint32_t)ref) & ~kLeafQuadBit, first = code >> 2, cnt = (code & 3u) + 1u;
                    for (uint32_t t = 0; t < cnt; ++t) {
                        const float *v = d->tri_verts + 9 * (size_t)hs.slot_orig[first + t];
                        // a leaf box holds the triangle, or — for a PIECE of a large triangle (early split clipping) — at
                        // least overlaps its box
                        double tl[3] = { 1e300, 1e300, 1e300 }, th[3] = { -1e300, -1e300, -1e300 };
                        for (int p = 0; p < 3; ++p) for (int a = 0; a < 3; ++a) { tl[a] = std::min(tl[a], (double)v[3 * p + a]); th[a] = std::max(th[a], (double)v[3 * p + a]); }
                        for (int a = 0; a < 3; ++a)
                            if (th[a] < lo[a] || tl[a] > hi[a]) return -21;      // the leaf box does not even touch its triangle
                    }
                }
            }
        }
    }
    return 0;
}
