// mtr_api.hip — the C-ABI of include/mitransient_amd.h.
//
// Host side of the library: scene ingestion (derived per-triangle frames, emitter normals,
// BVH2 build, upload), render planning and launches, film develop/clear, the stand-alone
// scatter-add.  There is no CPU execution path: without a HIP device every entry point fails.
#include "../../include/mitransient_amd.h"
#include "mtr_knobs.h"
#include "mtr_scene_host.h"
#include "mtr_core.h"
#include "mtr_kernels.h"

#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <string>
#include <vector>
#include <algorithm>
#include <utility>

using namespace mtr;

struct mtr_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    int n_cu = 256;
    std::string err;
    DevCounters *d_counters = nullptr;
    uint32_t *d_ticket = nullptr;          // work-ticket counters: [0, 16) k_fused launches in rotation (launches of consecutive
                                           // row bands may overlap on two streams: each needs its own), [16, 18) wavefront segments
    uint32_t fused_launches = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t ev2 = nullptr, ev3 = nullptr;               // mtr_splat_add: the partitioned passes, timed apart from the first pass (the workspace allocation in between is host time)
    float *d_freq = nullptr; uint32_t freq_cap = 0;      // phasor film frequencies of a ctx-level call (mtr_splat_add)
    void *d_runs = nullptr; size_t runs_cap = 0;         // mtr_splat_add variant 1: sortedness flag + run table
    uint32_t *d_band_count = nullptr; uint32_t band_cap = 0;     // mtr_render_params.n_bands: flushed pixels per band of the launch in flight
    void *d_part = nullptr; size_t part_cap = 0;         // ... and the partition workspace of unsorted input (at most 256 MiB of it kept between calls, until mtr_ctx_trim / destroy)
};

struct WfWorkspace {            // MTR_MODE_WAVEFRONT buffers, sized for one tile, reused across renders
    void *planes = nullptr, *q_live = nullptr, *q_ray = nullptr, *q_mat = nullptr, *q_shadow = nullptr, *r_shadow = nullptr, *occ = nullptr, *counts = nullptr, *rec = nullptr, *rec_count = nullptr, *q_zombie = nullptr;
    uint32_t n_slots = 0, P = 0, rec_cap = 0, rows = 0;
    uint32_t *host_count = nullptr;       // pinned: live counts read back between bounce chunks (two words, alternating)
    hipEvent_t poll_ev[2] = { nullptr, nullptr };     // ... and the events that say a word has landed
};

struct NlosDev {                // NLOS tier: device tables + constants (mtr_scene_set_nlos)
    bool on = false;
    NlosConst k{};
    void *shapes = nullptr, *tables = nullptr, *hg_tris = nullptr, *hg_vn = nullptr, *targets = nullptr;
    std::vector<mtr_shape> host_shapes;      // kept: the triangle -> shape table of the scene
};

struct mtr_scene {
    mtr_ctx *ctx = nullptr;
    WfWorkspace wf;
    NlosDev nlos;
    std::vector<float> tri_verts;            // host copy (NLOS tables are re-derived when the laser moves)
    std::vector<float> tri_normals;          // ... and the vertex normals (empty without): Mesh::sample_position on hidden meshes
    float bb_lo[3] = { 0, 0, 0 }, bb_hi[3] = { 0, 0, 0 };   // bounds of the triangles (the grid of the wavefront organisation's trace order)
    uint32_t n_emitters_area = 0;
    bool grey_scene = false;                                 // kTrGrey without the NLOS laser (mtr_scene_set_nlos decides with it)
    SceneDev dev{};
    Camera cam{};
    Film film{};
    mtr_film_desc film_desc{};
    float *d_freq = nullptr;               // phasor film: device copy of film_desc.frequencies
    std::vector<float> h_freq;
    uint32_t n_leaves = 0;
    std::vector<void *> allocs;
    SplatLog log{ nullptr, 0, nullptr };
};

static thread_local std::string g_err;

static int fail(mtr_ctx *c, int code, const std::string &msg)
{
    if (c) c->err = msg; else g_err = msg;
    return code;
}
// CUs the persistent fused kernel may occupy (mtr_render_params.reserve_cus: the rest stays free for other streams' kernels)
static int usable_cus(const mtr_ctx *c, const mtr_render_params *p)
{
    const int keep = (int)std::min<uint32_t>(p->reserve_cus, (uint32_t)(c->n_cu > 1 ? c->n_cu - 1 : 0));
    return c->n_cu - keep;
}

#define HIP_TRY(c, expr)                                                                       \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail((c), e_ == hipErrorOutOfMemory ? MTR_ERR_OOM : MTR_ERR_HIP,             \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                     \
    } while (0)

extern "C" {

int mtr_abi_version(void) { return MTR_ABI_VERSION; }

const char *mtr_last_error(const mtr_ctx *c) { return c ? c->err.c_str() : g_err.c_str(); }

int mtr_ctx_create(int device_ordinal, mtr_ctx **out)
{
    if (!out) return fail(nullptr, MTR_ERR_INVALID, "mtr_ctx_create: out is NULL");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0)
        return fail(nullptr, MTR_ERR_NO_DEVICE, "no HIP device visible; this library has no CPU path");
    if (device_ordinal < 0 || device_ordinal >= n)
        return fail(nullptr, MTR_ERR_INVALID, "mtr_ctx_create: device ordinal out of range");
    mtr_ctx *c = new mtr_ctx();
    c->device = device_ordinal;
    HIP_TRY(nullptr, hipSetDevice(device_ordinal));
    hipDeviceProp_t prop;
    HIP_TRY(nullptr, hipGetDeviceProperties(&prop, device_ordinal));
    c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    HIP_TRY(nullptr, hipMalloc((void **)&c->d_counters, sizeof(DevCounters)));
    HIP_TRY(nullptr, hipMalloc((void **)&c->d_ticket, 32 * sizeof(uint32_t)));
    HIP_TRY(nullptr, hipEventCreate(&c->ev0));
    HIP_TRY(nullptr, hipEventCreate(&c->ev1));
    HIP_TRY(nullptr, hipEventCreate(&c->ev2));
    HIP_TRY(nullptr, hipEventCreate(&c->ev3));
    *out = c;
    return MTR_OK;
}

void mtr_ctx_destroy(mtr_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->d_counters) (void)hipFree(c->d_counters);
    if (c->d_ticket) (void)hipFree(c->d_ticket);
    if (c->d_freq) (void)hipFree(c->d_freq);
    if (c->d_runs) (void)hipFree(c->d_runs);
    if (c->d_part) (void)hipFree(c->d_part);
    if (c->d_band_count) (void)hipFree(c->d_band_count);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->ev2) (void)hipEventDestroy(c->ev2);
    if (c->ev3) (void)hipEventDestroy(c->ev3);
    delete c;
}

int mtr_ctx_set_stream(mtr_ctx *c, void *s)
{
    if (!c) return fail(nullptr, MTR_ERR_INVALID, "mtr_ctx_set_stream: ctx is NULL");
    c->stream = (hipStream_t)s;
    return MTR_OK;
}

} // extern "C"

static int check_film(mtr_ctx *c, const mtr_film_desc &d)
{
    if (d.width == 0 || d.height == 0 || d.temporal_bins == 0)
        return fail(c, MTR_ERR_INVALID, "film: width, height and temporal_bins must be positive");
    if (d.crop_width == 0 || d.crop_height == 0 || d.crop_offset_x + d.crop_width > d.width ||
        d.crop_offset_y + d.crop_height > d.height)
        return fail(c, MTR_ERR_INVALID, "film: invalid crop window");
    if (!(d.bin_width_opl > 0.0f)) return fail(c, MTR_ERR_INVALID, "film: bin_width_opl must be > 0");
    if (d.n_frequencies && !d.frequencies) return fail(c, MTR_ERR_INVALID, "film: n_frequencies > 0 but frequencies is NULL");
    if (d.n_frequencies && (d.laser_scan_width || d.laser_scan_height))
        return fail(c, MTR_ERR_INVALID, "film: a phasor film cannot be an exhaustive_scan film");
    if (d.n_frequencies > (1u << 20)) return fail(c, MTR_ERR_INVALID, "film: too many frequencies");
    if ((d.laser_scan_width == 0) != (d.laser_scan_height == 0))
        return fail(c, MTR_ERR_INVALID, "film: laser_scan_width and laser_scan_height must both be set (exhaustive_scan) or both be 0");
    if ((uint64_t)d.temporal_bins * (d.laser_scan_width ? d.laser_scan_width : 1u) * (d.laser_scan_height ? d.laser_scan_height : 1u) > 0x7fffffffull)
        return fail(c, MTR_ERR_INVALID, "film: laser_scan_width * laser_scan_height * temporal_bins exceeds 2^31");
    return MTR_OK;
}

template <class T>
static int upload(mtr_scene *s, const std::vector<T> &v, const T **out)
{
    size_t bytes = ((v.size() * sizeof(T) + 15) / 16) * 16;
    if (bytes == 0) bytes = 16;
    void *p = nullptr;
    HIP_TRY(s->ctx, hipMalloc(&p, bytes));
    s->allocs.push_back(p);
    HIP_TRY(s->ctx, hipMemset(p, 0, bytes));
    if (!v.empty()) HIP_TRY(s->ctx, hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = (const T *)p;
    return MTR_OK;
}

extern "C" {

static int scene_take_film(mtr_scene *s, const mtr_film_desc &f);

int mtr_scene_create(mtr_ctx *c, const mtr_scene_desc *d, mtr_scene **out)
{
    if (!c || !d || !out) return fail(c, MTR_ERR_INVALID, "mtr_scene_create: NULL argument");
    *out = nullptr;
    int rc = check_film(c, d->film);
    if (rc) return rc;
    HostScene hs;
    if (const char *msg = derive_scene(*d, hs)) return fail(c, MTR_ERR_INVALID, std::string("mtr_scene_create: ") + msg);
    HIP_TRY(c, hipSetDevice(c->device));

    mtr_scene *s = new mtr_scene();
    s->ctx = c;
    s->cam = hs.cam;
    rc = scene_take_film(s, d->film);
    if (rc) { delete s; return rc; }

#define UP(vec, field)                                                       \
    do { rc = upload(s, vec, &s->dev.field); if (rc) { mtr_scene_destroy(s); return rc; } } while (0)
    UP(hs.nodes, nodes); UP(hs.tpairs, tpairs); UP(hs.tshade, tshade); UP(hs.mats, mats); UP(hs.ems, ems);
    s->dev.samp_tris = nullptr; s->dev.samp_vn = nullptr; s->dev.face_pmf = s->dev.face_cdf = nullptr; s->dev.vnormals = nullptr; s->dev.texels = s->dev.tex_info = s->dev.uvs = nullptr;
    s->dev.wnodes = nullptr; s->dev.n_wnodes = (uint32_t)hs.wnodes.size();
    if (hs.has_wide) UP(hs.wnodes, wnodes);
    s->dev.wnodes4 = nullptr; s->dev.n_wnodes4 = (uint32_t)hs.wnodes4.size();
    UP(hs.wnodes4, wnodes4);
    s->dev.wnodes8q = nullptr; s->dev.n_wnodes8q = (uint32_t)hs.wnodes8q.size();
    if (!hs.wnodes8q.empty()) UP(hs.wnodes8q, wnodes8q);
    if (!hs.samp_tris.empty()) { UP(hs.samp_tris, samp_tris); UP(hs.face_pmf, face_pmf); UP(hs.face_cdf, face_cdf); }
    if (!hs.samp_vn.empty()) UP(hs.samp_vn, samp_vn);
    if (!hs.vnormals.empty()) UP(hs.vnormals, vnormals);
    if (!hs.texels.empty()) { UP(hs.texels, texels); UP(hs.tex_info, tex_info); UP(hs.uvs, uvs); }
#undef UP
    s->dev.n_nodes = (uint32_t)hs.nodes.size(); s->dev.n_slots = (uint32_t)hs.tshade.size();
    s->dev.n_mats = d->n_materials; s->dev.n_ems = d->n_emitters;
    s->dev.has_rough = 0u;
    for (uint32_t i = 0; i < d->n_materials; ++i)
        if (bsdf_is_rough(d->materials[i].type) || d->materials[i].type == MTR_BSDF_THINDIELECTRIC) s->dev.has_rough = 1u;
    if (!hs.vnormals.empty() || !hs.texels.empty()) s->dev.has_rough = 1u;      // smooth-shaded triangles, bitmap textures: the extended shading code as well
    // scene traits (mtr_core.h): facts about the tables that let the kernels drop shading code no hit can reach
    s->dev.traits = 0u;
    if (!s->dev.has_rough) {
        bool diffuse_only = d->n_materials > 0;
        for (uint32_t i = 0; i < d->n_materials; ++i)
            if (d->materials[i].type != MTR_BSDF_DIFFUSE || (d->materials[i].flags & MTR_MAT_TWOSIDED)) diffuse_only = false;
        if (diffuse_only) s->dev.traits |= kTrDiffuse;
        if (d->n_emitters == 1 && !hs.ems[0].is_mesh) s->dev.traits |= kTrOneRectEmitter;
        bool leaf_pairs = hs.has_wide;
        for (const WNode &n : hs.wnodes)
            for (uint32_t k = 0; k < n.count; ++k) {
                const uint32_t code = ~(uint32_t)n.ref[k];
                if (n.ref[k] < 0 && !(code & kLeafQuadBit) && (code & 3u) + 1u > 2u) leaf_pairs = false;
            }
        if (leaf_pairs) s->dev.traits |= kTrLeafPair;
    }
    if (s->dev.has_rough) {          // kTrNoLobes: the extended shading code is needed for normals / bitmaps only
        bool lobes = false;
        for (uint32_t i = 0; i < d->n_materials; ++i)
            if (bsdf_is_rough(d->materials[i].type) || d->materials[i].type == MTR_BSDF_THINDIELECTRIC) lobes = true;
        if (!lobes) s->dev.traits |= kTrNoLobes;
    }
    {   // kTrGrey (the scene's part; mtr_scene_set_nlos adds the laser's): three equal channels in every colour, no bitmap
        auto eq3 = [](const float *v) { return memcmp(v, v + 1, sizeof(float)) == 0 && memcmp(v, v + 2, sizeof(float)) == 0; };
        bool grey = hs.texels.empty();
        for (uint32_t i = 0; grey && i < d->n_materials; ++i) {
            const mtr_material &m = d->materials[i];
            const bool aniso = (m.flags & MTR_MAT_ANISOTROPIC) != 0u;
            grey = m.albedo_texture == 0u && eq3(m.a) && eq3(m.c) &&
                   ((aniso && m.type == MTR_BSDF_ROUGHDIELECTRIC) || eq3(m.b)) && ((aniso && m.type == MTR_BSDF_ROUGHCONDUCTOR) || eq3(m.c2));
        }
        for (uint32_t i = 0; grey && i < d->n_emitters; ++i) grey = eq3(d->emitters[i].radiance);
        s->grey_scene = grey;
        if (grey) s->dev.traits |= kTrGrey;
    }
    // kTrFlatTop (any materials): the root's children are rectangles and box nodes, the boxes' nodes follow the root in order
    memset(&s->dev.flat, 0, sizeof s->dev.flat);
    if (hs.has_wide && !hs.wnodes.empty() && hs.wide_levels <= 2 && !mtr::knob("MTR_NO_FLAT")) {
        const WNode &root = hs.wnodes[0];
        bool flat = root.flags == 0u && root.count >= 1u;
        uint32_t n_inner = 0u, prim_mask = 0u;
        for (uint32_t k = 0; flat && k < root.count; ++k) {
            const int32_t ref = root.ref[k];
            if (k < root.n_quads) { flat = ref < 0 && ((~(uint32_t)ref) & kLeafQuadBit) != 0u; prim_mask |= 1u << k; }
            else if (ref < 0) { flat = ((~(uint32_t)ref) & kLeafQuadBit) == 0u; prim_mask |= 1u << k; }             // a triangle leaf
            else {            // an inner child: a box node, and the n-th of them is node n + 1
                flat = ref == (int32_t)(1u + n_inner) && (size_t)ref < hs.wnodes.size() && hs.wnodes[ref].flags == 3u && hs.wnodes[ref].count == 6u;
                ++n_inner;
            }
        }
        flat = flat && n_inner <= kFlatMaxBoxes;
        if (flat) {
            FlatTop &ft = s->dev.flat;
            ft.n_quads = root.n_quads; ft.n_boxes = n_inner; ft.node0 = 1u; ft.prim_mask = prim_mask;
            for (uint32_t b = 0; b < ft.n_boxes; ++b) {
                const float *x = hs.wnodes[1u + b].xf;
                memcpy(ft.xf[b], x, 12 * sizeof(float));
                for (int k = 0; k < 3; ++k) ft.xf[b][12 + k] = (fabsf(x[4 * k]) + fabsf(x[4 * k + 1]) + fabsf(x[4 * k + 2])) * 1.000001f;      // S: row sums of |R| (rounded up)
                ft.xf[b][15] = 0.0f;
            }
            s->dev.traits |= kTrFlatTop;
            if (prim_mask >> root.n_quads) s->dev.traits |= kTrFlatLeaves;
        }
    }
    s->dev.bvh_depth = hs.bvh_depth; s->n_leaves = hs.n_leaves;
    s->dev.wide_levels = hs.wide_levels; s->dev.wide4_levels = hs.wide4_levels; s->dev.wide8q_levels = hs.wide8q_levels;
    s->tri_verts.assign(d->tri_verts, d->tri_verts + 9 * (size_t)d->n_tris);
    for (int k = 0; k < 3; ++k) { s->bb_lo[k] = d->n_tris ? INFINITY : 0.0f; s->bb_hi[k] = d->n_tris ? -INFINITY : 0.0f; }
    for (size_t i = 0; i < 3 * (size_t)d->n_tris; ++i)
        for (int k = 0; k < 3; ++k) { const float v = d->tri_verts[3 * i + k]; s->bb_lo[k] = std::min(s->bb_lo[k], v); s->bb_hi[k] = std::max(s->bb_hi[k], v); }
    if (d->tri_normals) s->tri_normals.assign(d->tri_normals, d->tri_normals + 9 * (size_t)d->n_tris);
    s->n_emitters_area = d->n_emitters;
    if (d->nlos) {
        rc = mtr_scene_set_nlos(s, d->nlos);
        if (rc) { mtr_scene_destroy(s); return rc; }
    }
    *out = s;
    return MTR_OK;
}

int mtr_scene_set_nlos(mtr_scene *s, const mtr_nlos_desc *n)
{
    if (!s || !n) return fail(s ? s->ctx : nullptr, MTR_ERR_INVALID, "mtr_scene_set_nlos: NULL argument");
    mtr_ctx *c = s->ctx;
    HIP_TRY(c, hipSetDevice(c->device));
    mtr_scene_desc d{};
    d.n_tris = (uint32_t)(s->tri_verts.size() / 9); d.tri_verts = s->tri_verts.data(); d.n_emitters = s->n_emitters_area;
    d.tri_normals = s->tri_normals.empty() ? nullptr : s->tri_normals.data();
    d.film = s->film_desc;
    memcpy(d.camera.sample_to_camera, s->cam.s2c, sizeof s->cam.s2c);
    memcpy(d.camera.to_world, s->cam.tw, sizeof s->cam.tw);
    d.camera.near_clip = s->cam.near_clip; d.camera.far_clip = s->cam.far_clip;
    d.nlos = n;
    HostNlos hn;
    if (const char *msg = derive_nlos(d, hn)) return fail(c, MTR_ERR_INVALID, std::string("mtr_scene_set_nlos: ") + msg);
    NlosDev &D = s->nlos;
    void **old[] = { &D.shapes, &D.tables, &D.hg_tris, &D.hg_vn, &D.targets };
    for (void **p : old) if (*p) { (void)hipFree(*p); *p = nullptr; }
    const size_t ns = hn.shapes.size(), nt = hn.face_pmf.size();
    HIP_TRY(c, hipMalloc(&D.shapes, ns * sizeof(NlosShape)));
    HIP_TRY(c, hipMemcpy(D.shapes, hn.shapes.data(), ns * sizeof(NlosShape), hipMemcpyHostToDevice));
    std::vector<float> tab;                                       // shape_pmf | shape_cdf | face_pmf | face_cdf
    tab.insert(tab.end(), hn.shape_pmf.begin(), hn.shape_pmf.end());
    tab.insert(tab.end(), hn.shape_cdf.begin(), hn.shape_cdf.end());
    tab.insert(tab.end(), hn.face_pmf.begin(), hn.face_pmf.end());
    tab.insert(tab.end(), hn.face_cdf.begin(), hn.face_cdf.end());
    HIP_TRY(c, hipMalloc(&D.tables, tab.size() * 4));
    HIP_TRY(c, hipMemcpy(D.tables, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMalloc(&D.hg_tris, hn.hg_tris.size() * sizeof(q4)));
    HIP_TRY(c, hipMemcpy(D.hg_tris, hn.hg_tris.data(), hn.hg_tris.size() * sizeof(q4), hipMemcpyHostToDevice));
    if (!hn.hg_vn.empty()) {
        HIP_TRY(c, hipMalloc(&D.hg_vn, hn.hg_vn.size() * sizeof(q4)));
        HIP_TRY(c, hipMemcpy(D.hg_vn, hn.hg_vn.data(), hn.hg_vn.size() * sizeof(q4), hipMemcpyHostToDevice));
    }
    const size_t n_targets = nlos_target_count(hn.k);
    HIP_TRY(c, hipMalloc(&D.targets, n_targets * sizeof(q4)));
    D.k = hn.k;
    D.k.shapes = (const NlosShape *)D.shapes;
    D.k.shape_pmf = (const float *)D.tables; D.k.shape_cdf = D.k.shape_pmf + ns;
    D.k.face_pmf = D.k.shape_cdf + ns; D.k.face_cdf = D.k.face_pmf + nt;
    D.k.hg_tris = (const q4 *)D.hg_tris; D.k.hg_vn = (const q4 *)D.hg_vn; D.k.targets = (const q4 *)D.targets;
    HIP_TRY(c, launch_nlos_prepare(s->dev, D.k, (q4 *)D.targets, c->stream));    // scanned points + laser axis hit
    D.on = true;
    // kTrGrey: ... and the laser's irradiance
    const bool laser_grey = memcmp(n->laser_irradiance, n->laser_irradiance + 1, sizeof(float)) == 0 && memcmp(n->laser_irradiance, n->laser_irradiance + 2, sizeof(float)) == 0;
    if (s->grey_scene && laser_grey) s->dev.traits |= kTrGrey; else s->dev.traits &= ~kTrGrey;
    return MTR_OK;
}

void mtr_scene_destroy(mtr_scene *s)
{
    if (!s) return;
    if (s->ctx) (void)hipSetDevice(s->ctx->device);
    for (void *p : s->allocs) (void)hipFree(p);
    void *w[] = { s->wf.planes, s->wf.q_live, s->wf.q_ray, s->wf.q_mat, s->wf.q_shadow, s->wf.r_shadow, s->wf.occ, s->wf.counts, s->wf.rec, s->wf.rec_count, s->wf.q_zombie };
    for (void *p : w) if (p) (void)hipFree(p);
    if (s->wf.host_count) (void)hipHostFree(s->wf.host_count);
    for (hipEvent_t e : s->wf.poll_ev) if (e) (void)hipEventDestroy(e);
    void *nl[] = { s->nlos.shapes, s->nlos.tables, s->nlos.hg_tris, s->nlos.hg_vn, s->nlos.targets, s->d_freq };
    for (void *p : nl) if (p) (void)hipFree(p);
    delete s;
}

// the scene keeps its own host + device copy of a phasor film's frequencies (the caller's array need not outlive the call)
static int scene_take_film(mtr_scene *s, const mtr_film_desc &f)
{
    mtr_ctx *c = s->ctx;
    s->film = film_from_desc(f);
    s->film_desc = f;
    s->film.freq = nullptr; s->film_desc.frequencies = nullptr;
    if (f.n_frequencies) {
        const std::vector<float> nf(f.frequencies, f.frequencies + f.n_frequencies);
        if (nf != s->h_freq || !s->d_freq) {
            HIP_TRY(c, hipSetDevice(c->device));
            if (s->d_freq) { HIP_TRY(c, hipStreamSynchronize(c->stream)); (void)hipFree(s->d_freq); s->d_freq = nullptr; }
            HIP_TRY(c, hipMalloc((void **)&s->d_freq, nf.size() * 4));
            HIP_TRY(c, hipMemcpy(s->d_freq, nf.data(), nf.size() * 4, hipMemcpyHostToDevice));
            s->h_freq = nf;
        }
        s->film.freq = s->d_freq; s->film_desc.frequencies = s->h_freq.data();
    }
    return MTR_OK;
}

int mtr_scene_set_film(mtr_scene *s, const mtr_film_desc *f)
{
    if (!s || !f) return fail(s ? s->ctx : nullptr, MTR_ERR_INVALID, "mtr_scene_set_film: NULL argument");
    int rc = check_film(s->ctx, *f);
    if (rc) return rc;
    return scene_take_film(s, *f);
}

int mtr_scene_bvh_info(const mtr_scene *s, uint32_t *n_nodes, uint32_t *max_depth, uint32_t *n_leaves)
{
    if (!s) return MTR_ERR_INVALID;
    if (n_nodes) *n_nodes = s->dev.n_nodes;
    if (max_depth) *max_depth = s->dev.bvh_depth;
    if (n_leaves) *n_leaves = s->n_leaves;
    return MTR_OK;
}

int mtr_scene_traits(const mtr_scene *s, uint32_t *traits)
{
    if (!s || !traits) return MTR_ERR_INVALID;
    static_assert(MTR_TRAIT_DIFFUSE == kTrDiffuse && MTR_TRAIT_ONE_RECT_EMITTER == kTrOneRectEmitter && MTR_TRAIT_LEAF_PAIR == kTrLeafPair &&
                  MTR_TRAIT_FLAT_TOP == kTrFlatTop && MTR_TRAIT_FLAT_LEAVES == kTrFlatLeaves && MTR_TRAIT_NO_LOBES == kTrNoLobes &&
                  MTR_TRAIT_GREY == kTrGrey, "public trait bits");
    *traits = s->dev.traits;
    return MTR_OK;
}

} // extern "C"

// ---- MTR_MODE_WAVEFRONT: host loop over tiles and bounces ------------------------------------
static int wf_alloc(mtr_scene *s, uint32_t n_slots, uint32_t P, uint32_t n_seg, uint32_t rec_cap)
{
    mtr_ctx *c = s->ctx;
    WfWorkspace &w = s->wf;
    if (w.n_slots >= n_slots && w.P >= P && w.rec_cap == rec_cap && w.rows >= n_seg && w.planes) return MTR_OK;
    void **ptrs[] = { &w.planes, &w.q_live, &w.q_ray, &w.q_mat, &w.q_shadow, &w.r_shadow, &w.occ, &w.counts, &w.rec, &w.rec_count, &w.q_zombie };
    w.n_slots = 0; w.P = 0; w.rec_cap = 0; w.rows = 0;          // sizes are valid only once every buffer below exists
    for (void **p : ptrs) if (*p) { (void)hipFree(*p); *p = nullptr; }
    HIP_TRY(c, hipMalloc(&w.planes, wf_planes_bytes(n_slots)));
    HIP_TRY(c, hipMalloc(&w.q_live, (size_t)2 * n_slots * 4));
    HIP_TRY(c, hipMalloc(&w.q_ray, (size_t)2 * n_slots * 32));                       // rays of the live lists, in list order
    HIP_TRY(c, hipMalloc(&w.q_mat, (size_t)kWfKeys * n_slots * 4));
    HIP_TRY(c, hipMalloc(&w.q_shadow, (size_t)2 * n_slots * 2 + 64));                // TRACE ORDER: q_order | q_order_sh, 16 bits per list position (the former shadow-slot list's buffer)
    HIP_TRY(c, hipMalloc(&w.q_zombie, (size_t)2 * n_slots * 4));                    // paths that ended with an emitter-sampling term parked, per parity
    HIP_TRY(c, hipMalloc(&w.r_shadow, (size_t)n_slots * 32));
    HIP_TRY(c, hipMalloc(&w.occ, (size_t)n_slots + (size_t)n_seg * 16u + 16u));       // [n_seg][seg rounded up to 16] occlusion flags in shadow-list order
    HIP_TRY(c, hipMalloc(&w.counts, ((size_t)n_seg * (7 + kWfKeys) + 16) * 4));     // seg_live[2][n_seg], seg_mat[n_seg][5], seg_shadow[n_seg], live_total + seg_list_n[2] (16 words), seg_list[2][n_seg], seg_zombie[2][n_seg]
    HIP_TRY(c, hipMalloc(&w.rec, std::max<size_t>(16, (size_t)P * rec_cap * 16)));
    HIP_TRY(c, hipMalloc(&w.rec_count, (size_t)P * 4));
    if (!w.host_count) HIP_TRY(c, hipHostMalloc((void **)&w.host_count, 64));
    for (hipEvent_t &e : w.poll_ev) if (!e) HIP_TRY(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    w.n_slots = n_slots; w.P = P; w.rec_cap = rec_cap; w.rows = n_seg;
    return MTR_OK;
}

static int wf_render(mtr_scene *s, const mtr_render_params *p, float *t4, float *s4, const RenderConst &rc,
                     float *trace_ms, float *scatter_ms, uint32_t *n_trace, uint32_t *n_scatter, bool timed, uint32_t *n_trace_kernel,
                     bool may_block, float *shade_ms)
{
    mtr_ctx *c = s->ctx;
    const Film &f = s->film;
    WfConfig cfg{};
    if (!wf_plan(s->dev, cfg)) return fail(c, MTR_ERR_UNSUPPORTED, "wavefront: BVH too deep for the LDS stack");
    const uint32_t n_pixels = p->pixel_end - p->pixel_begin;
    const uint32_t spp_chunk = p->spp_end - p->spp_begin;
    // tile = P pixels x S samples (2^25 slots); segment = G whole pixels (about 4096 slots: with the persistent
    // k_wf_trace a segment is drained once per launch, so longer segments waste less — staircase 1024: 425 ms,
    // 2048: 390, 4096: 360, 8192: 397)
    // Tile: as many slots as half of the free device memory holds, at most 2^28 (92 GB of workspace at 341 B per slot).  Every
    // bounce of every tile costs four launches with ~0.1 ms of fixed cost each, and with max_depth 65 most of them run nearly
    // empty: config 5 (2^29 slots) with tiles of 2^25 / 2^26 / 2^27 / 2^28 slots: 2.01 / 1.80 / 1.70 / 1.59 s per render
    // (config 2 in this organisation: 2^22 269 ms, 2^24 174 ms, 2^25 168 ms).
    // Segment: scenes walked in HBM 8192 slots (config 5: 2048 / 4096 / 8192 / 16384 slots: 338 / 288 / 275 / 273 ms at 256 spp
    // with the 8-wide tree); scenes staged in LDS the same since k_wf_shade's state diet (round 5; config 2 with 2048 / 4096 / 6144 /
    // 8192 / 12288 / 16384 slots: 97.8 / 81.9 / 79.4 / 77.1 - 79.7 / 77.5 / 80.8 ms, 108 triangles 131.9 / 117.2 / 113.9 / 116.1 / 114.7 /
    // 123.4 ms; rounds 2-4 had 4096: 141 against 169 ms with 8192 then).
    uint32_t kTileSlots = 1u << 28; uint32_t kSegSlots = 8192u;
    {
        size_t free_b = 0, total_b = 0;
        const size_t per_slot = 344;                                   // planes 128 + queues 52 + rays 96 + records 64 + occlusion 1, rounded up
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            // half of what is free now (the workspace this scene already holds counts as free), and never more than a third of
            // the device: the caller's allocator (films, all-gather buffers, a second scene) needs room the driver cannot see
            size_t budget = (free_b + (size_t)s->wf.n_slots * per_slot) / 2;
            if (budget > total_b / 3) budget = total_b / 3;
            while (kTileSlots > (1u << 22) && (size_t)kTileSlots * per_slot > budget) kTileSlots >>= 1;
        } else kTileSlots = 1u << 25;
    }
    if (const char *e = mtr::knob("MTR_WF_TILE_LOG2")) kTileSlots = 1u << atoi(e);      // experiments
    if (const char *e = mtr::knob("MTR_WF_SEG")) kSegSlots = (uint32_t)atoi(e);
    if (kSegSlots > 32768u) kSegSlots = 32768u;          // a segment holds at most 2^16 slots (seg < kSegSlots + S): k_wf_trace packs (list position, slot) into one word
    const uint32_t S = spp_chunk < 4096u ? spp_chunk : 4096u;
    // (at most 1024 pixels per segment: k_wf_shade keeps 20 B of LDS per pixel of its segment — record-list tail, steady sums —
    // and a render of very few samples per pixel would otherwise ask for more LDS than a CU has: 8192 pixels = 164 KB)
    const uint32_t G = std::min<uint32_t>((kSegSlots + S - 1) / S, 1024u);
    uint32_t P = kTileSlots / S; if (P < G) P = G; if (P > n_pixels) P = n_pixels;
    const uint32_t n_slots_max = P * S;
    const uint32_t seg = G * S;
    const uint32_t n_seg_max = (P + G - 1) / G;      // (of the first attempt; wf_alloc below may settle for a smaller tile)
    // time-bin records: per-pixel lists sized for 4 contributions per path; the rest (and rows that do not
    // fit LDS) fall back to f32 atomics on the film
    const bool rows_fit = (size_t)f.bins * 12u <= 150u * 1024u;
    const uint32_t rec_cap = rows_fit ? S * 4u : 0u;
    int rc_ = wf_alloc(s, n_slots_max, P, n_seg_max, rec_cap);
    // out of memory (someone else took it between hipMemGetInfo and here): halve the tile until the workspace fits
    while (rc_ == MTR_ERR_OOM && P > G && (size_t)P * S > (1u << 22)) {
        (void)hipGetLastError();                       // (the failed hipMalloc must not surface at the next launch check)
        P = std::max(G, ((P / 2 + G - 1) / G) * G);
        rc_ = wf_alloc(s, P * S, P, (P + G - 1) / G, rec_cap);
    }
    if (rc_) return rc_;
    WfWorkspace &w = s->wf;

    WfArgs a{};
    a.sc = s->dev; a.cam = s->cam; a.film = f; a.rc = rc;
    a.planes = (float *)w.planes; a.q_live = (uint32_t *)w.q_live; a.q_ray = (float4 *)w.q_ray; a.q_mat = (uint32_t *)w.q_mat;
    a.q_shadow = nullptr; a.r_shadow = (float4 *)w.r_shadow; a.occ = (uint8_t *)w.occ;
    // TRACE ORDER (mtr_kernels.h; an EXPERIMENT, off: MTR_WF_SORT=1 / MTR_WF_SORT_SH=1 in the experiments build): rays of the path
    // tier traced sorted by (origin cell, direction octant); the grid's 5 bits go to the axes along which the scene is longest.
    // Measured (round 6, config 5 at 256 spp): k_wf_trace 108.2 -> 110.3 ms, render 155.6 -> 160.1 ms; config 2 wavefront 81.4 -> 88.9 ms.
    a.q_order = nullptr; a.q_order_sh = nullptr;
    if (!s->nlos.on && mtr::knob("MTR_WF_SORT") && s->tri_verts.size() >= 9) {
        a.q_order = (uint16_t *)w.q_shadow;
        if (mtr::knob("MTR_WF_SORT_SH")) a.q_order_sh = (uint16_t *)w.q_shadow + w.n_slots;
        float ext[3]; uint32_t bits[3] = { 0u, 0u, 0u };
        for (int k = 0; k < 3; ++k) ext[k] = std::max(s->bb_hi[k] - s->bb_lo[k], 1e-20f);
        for (int b = 0; b < 5; ++b) {
            int m = 0;
            for (int k = 1; k < 3; ++k) if (ext[k] / (float)(1u << bits[k]) > ext[m] / (float)(1u << bits[m])) m = k;
            bits[m]++;
        }
        for (int k = 0; k < 3; ++k) { a.sort_lo[k] = s->bb_lo[k]; a.sort_scale[k] = (float)(1u << bits[k]) / ext[k]; a.sort_bits[k] = bits[k]; }
    }
    a.q_zombie = (uint32_t *)w.q_zombie;
    a.rec = (uint4 *)w.rec; a.rec_count = (uint32_t *)w.rec_count; a.rec_cap = rec_cap;
    a.film_out = t4; a.steady_out = s4; a.counters = c->d_counters; a.log = s->log;
    a.G = G; a.seg = seg;
    a.nlos_on = s->nlos.on ? 1u : 0u;
    if (s->nlos.on) a.nlos = s->nlos.k;
    const int grid_full = c->n_cu * 8;
    // NLOS paths end by the integrator's own rules (filter depth, roulette): the host polls the live count like an unbounded render
    // (deep bounded renders poll too: with max_depth 65 no path of config 5 is alive after some 35 bounces, and every bounce of
    // every tile is four launches)
    // ... but only in calls that block anyway (counters / timings requested): a caller that keeps mtr_render asynchronous —
    // row bands overlapped with collectives — is not stalled inside it; its empty bounces cost 4 us per launch (live segment list)
    const bool unbounded = p->max_depth < 0 || s->nlos.on || (p->max_depth > 16 && may_block);
    // the reference loop always runs its first iteration (emission of the camera-ray hit), also at max_depth 0
    const uint32_t max_depth = p->max_depth < 0 ? 0xffffffffu : (p->max_depth == 0 ? 1u : (uint32_t)p->max_depth + (s->nlos.on ? 2u : 0u));
    std::vector<std::pair<hipEvent_t, hipEvent_t>> scatter_ev, trace_ev, shade_ev;
    // (timed renders only) events around every k_wf_trace launch — the dominant kernel of scenes in HBM is timed alone — and
    // around the HBM-bound k_wf_shade (mtr_kernel_times.wf_shade_ms)
    auto launch_timed = [&](int which, int grid_, std::vector<std::pair<hipEvent_t, hipEvent_t>> &bucket) -> hipError_t {
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (timed) {
            hipError_t e = hipEventCreate(&e0); if (e != hipSuccess) return e;
            e = hipEventCreate(&e1); if (e != hipSuccess) return e;
            e = hipEventRecord(e0, c->stream); if (e != hipSuccess) return e;
        }
        hipError_t e = launch_wf(a, cfg, which, grid_, c->stream);
        if (e != hipSuccess) return e;
        if (timed) { e = hipEventRecord(e1, c->stream); if (e != hipSuccess) return e; bucket.push_back({ e0, e1 }); }
        return hipSuccess;
    };
    auto trace_timed = [&](int which, int grid_) -> hipError_t { return launch_timed(which, grid_, trace_ev); };

    for (uint32_t s0 = 0; s0 < spp_chunk; s0 += S) {
        const uint32_t Scur = std::min(S, spp_chunk - s0);
        for (uint32_t pix = 0; pix < n_pixels; pix += P) {
            const uint32_t Pcur = std::min(P, n_pixels - pix);
            a.pix0 = p->pixel_begin + pix; a.P = Pcur; a.spp_begin = p->spp_begin + s0; a.S = Scur;
            a.n_slots = Pcur * Scur;
            a.film_zero = ((p->flags & MTR_FLAG_FILM_ZERO) && s0 == 0 && rec_cap > 0) ? 1u : 0u;
            a.seg = G * Scur;                                       // segments always cover whole pixels
            a.occ_stride = (a.seg + 15u) & ~15u;
            a.n_seg = (Pcur + G - 1) / G;
            a.seg_live = (uint32_t *)w.counts;
            a.seg_mat = (uint32_t *)w.counts + (size_t)2 * a.n_seg;
            a.seg_shadow = (uint32_t *)w.counts + (size_t)(2 + kWfKeys) * a.n_seg;
            uint32_t *live_total = (uint32_t *)w.counts + (size_t)(3 + kWfKeys) * a.n_seg;
            a.live_total = unbounded ? live_total : nullptr;
            a.seg_list_n = live_total + 4;
            a.seg_list = live_total + 16;
            a.seg_zombie = a.seg_list + (size_t)2 * a.n_seg;
            const int grid = (int)std::min<uint32_t>(a.n_seg, (uint32_t)grid_full);
            const int grid_gen = (int)std::min<uint32_t>((a.n_slots + kBlock - 1) / kBlock, (uint32_t)grid_full);
            HIP_TRY(c, hipMemsetAsync(w.rec_count, 0, (size_t)Pcur * 4, c->stream));
            a.parity = 0;
            a.ticket = c->d_ticket + 16; a.ticket_cur = 0u;                          // segment tickets (k_wf_trace / shadow_gen / shade)
            HIP_TRY(c, hipMemsetAsync(a.ticket, 0, 2 * sizeof(uint32_t), c->stream));
            HIP_TRY(c, launch_wf(a, cfg, 0, grid_gen, c->stream));                   // raygen (writes live list 0)
            HIP_TRY(c, hipMemsetD32Async((hipDeviceptr_t)a.seg_list_n, (int)a.n_seg, 1, c->stream));     // bounce 0 walks every segment
            uint32_t depth = 0;
            // "Anyone left?" WITHOUT draining the stream (round 5).  Every 8 bounces the live count is copied to a pinned word and an
            // event is recorded behind the copy; the host then waits for the PREVIOUS poll's event — the count of 8 bounces ago —
            // while the chunk it has just enqueued keeps the GPU busy.  The loop therefore runs at most 8 bounces past the last
            // live path (launches over an empty segment list: 4 us each) and the stream never idles while the host decides; rounds
            // 1-4 synchronised the stream here, a bubble per chunk and a stall for a caller overlapping bands with collectives.
            uint32_t n_polls = 0;
            // (the count a poll reads is the one the PREVIOUS chunk of 8 bounces left: an unbounded render issues 8 - 16 bounce launches
            // over empty segment lists after its last path has died — about 4 us each; n_trace / trace_launches and the kernel times of
            // mtr_kernel_times include that speculative tail)
            auto poll_live = [&](bool &done) -> int {
                const uint32_t cur = n_polls & 1u;
                HIP_TRY(c, hipMemcpyAsync(w.host_count + cur, live_total, 4, hipMemcpyDeviceToHost, c->stream));
                HIP_TRY(c, hipEventRecord(w.poll_ev[cur], c->stream));
                if (n_polls) {
                    HIP_TRY(c, hipEventSynchronize(w.poll_ev[cur ^ 1u]));
                    if (w.host_count[cur ^ 1u] == 0u) done = true;
                }
                ++n_polls;
                return MTR_OK;
            };
            while (depth < max_depth) {
                if (unbounded) HIP_TRY(c, hipMemsetAsync(live_total, 0, 4, c->stream));
                HIP_TRY(c, hipMemsetAsync(a.seg_list_n + (a.parity ^ 1u), 0, 4, c->stream));     // the list this bounce's survivors build
                if (a.nlos_on) {          // NLOS tier: the whole loop iteration of transient_nlos_path in one launch per bounce
                    HIP_TRY(c, launch_wf(a, cfg, 5, grid, c->stream)); a.ticket_cur ^= 1u;
                    *n_trace += 1;
                    a.parity ^= 1u;
                    ++depth;
                    if (unbounded && (depth & 7u) == 0) { bool done = false; if (int r = poll_live(done)) return r; if (done) break; }
                    continue;
                }
                a.trace_any = 0u;
                a.first_bounce = depth == 0u ? 1u : 0u;
                HIP_TRY(c, trace_timed(1, grid)); a.ticket_cur ^= 1u;                // closest hit + material lists
                // shade: commits the emitter-sampling terms the previous bounce parked, runs the loop iteration once, writes the
                // shadow rays (scene in HBM) or traces them inline (scene in LDS), compacts the survivors
                HIP_TRY(c, launch_timed(2, grid, shade_ev)); a.ticket_cur ^= 1u;
                *n_trace += 2;
                if (!cfg.scene_lds) {                                                // scene in HBM/L2: the shadow rays get their own persistent trace
                    a.trace_any = 1u;
                    HIP_TRY(c, trace_timed(1, grid)); a.ticket_cur ^= 1u;            // occlusion of this bounce's shadow rays: read by the NEXT shade
                    *n_trace += 1;
                }
                a.parity ^= 1u;
                ++depth;
                if (unbounded && (depth & 7u) == 0) { bool done = false; if (int r = poll_live(done)) return r; if (done) break; }      // every 8 bounces: anyone left?
            }
            hipEvent_t a0 = nullptr, a1 = nullptr;
            if (timed) {
                HIP_TRY(c, hipEventCreate(&a0)); HIP_TRY(c, hipEventCreate(&a1));
                HIP_TRY(c, hipEventRecord(a0, c->stream));
            }
            HIP_TRY(c, launch_wf(a, cfg, 3, (int)std::min<uint32_t>(Pcur, (uint32_t)grid_full), c->stream));
            if (timed) { HIP_TRY(c, hipEventRecord(a1, c->stream)); scatter_ev.push_back({ a0, a1 }); }
            *n_scatter += 1;
        }
    }
    float acc_scatter = 0.0f;
    if (timed) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        for (auto &pr : scatter_ev) {
            float ms = 0.0f;
            HIP_TRY(c, hipEventElapsedTime(&ms, pr.first, pr.second));
            acc_scatter += ms;
            (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second);
        }
    }
    float acc_trace = 0.0f;
    for (auto &pr : trace_ev) {
        float ms = 0.0f;
        HIP_TRY(c, hipEventElapsedTime(&ms, pr.first, pr.second));
        acc_trace += ms;
        (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second);
    }
    float acc_shade = 0.0f;
    for (auto &pr : shade_ev) {
        float ms = 0.0f;
        HIP_TRY(c, hipEventElapsedTime(&ms, pr.first, pr.second));
        acc_shade += ms;
        (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second);
    }
    *scatter_ms = acc_scatter; *trace_ms = acc_trace; *n_trace_kernel = (uint32_t)trace_ev.size(); *shade_ms = acc_shade;
    return MTR_OK;
}

// MTR_MODE_AUTO -> the organisation that runs: the fused kernel when the whole scene can be staged in LDS (measured 143 vs
// 168 ms on config 2), the wavefront pipeline otherwise (BVH in HBM/L2: 21 vs 66 ms on an 81k-triangle scene)
static int resolve_mode(mtr_scene *s, const mtr_render_params *p, uint32_t n_pixels, uint32_t spp_chunk, uint32_t *mode_io,
                        uint32_t *developed_ok = nullptr)
{
    mtr_ctx *c = s->ctx;
    const Film &f = s->film;
    uint32_t mode = *mode_io;
    if (s->nlos.on && mode == MTR_MODE_AUTO)                             // (wavefront = the second organisation: on request, and for
        mode = (s->dev.has_rough && (p->flags & MTR_FLAG_DETERMINISTIC)) ? MTR_MODE_WAVEFRONT : MTR_MODE_FUSED;      // deterministic rows with the extended shading)
    if (f.n_freq) {                      // phasor film: (opl, value) records -> wavefront pipeline by default; LDS (Re, Im) rows in the fused kernel on request
        if (s->nlos.on) return fail(c, MTR_ERR_UNSUPPORTED, "mtr_render: phasor_hdr_film is not available for the NLOS tier");
        if (mode == MTR_MODE_AUTO) mode = MTR_MODE_WAVEFRONT;
    }
    if (s->dev.has_rough) {              // GGX lobes / smooth normals / bitmaps: f32 rows only (fused), or the wavefront pipeline
        const bool fused_ok = !f.n_freq && !(p->flags & MTR_FLAG_DETERMINISTIC);
        if (mode == MTR_MODE_FUSED && !fused_ok)
            return fail(c, MTR_ERR_UNSUPPORTED, "mtr_render: rough BSDFs / smooth-shaded triangles with a phasor film or deterministic rows need the wavefront mode");
        if (mode == MTR_MODE_AUTO && !fused_ok) mode = MTR_MODE_WAVEFRONT;
    }
    if (mode == MTR_MODE_AUTO) {
        FusedArgs probe{}; FusedConfig pc{};
        probe.sc = s->dev; probe.cam = s->cam; probe.film = f; probe.rc = make_render_const(*p, f, s->dev.n_ems); probe.nlos_on = s->nlos.on ? 1u : 0u;
        const bool fits = fused_plan(s->dev, f, n_pixels, spp_chunk, usable_cus(c, p), probe, pc) && pc.scene_lds;
        // ... and only a SHALLOW tree (a room of rectangles and a few objects: root + object nodes).  k_fused walks in lock-step — every
        // traversal costs its wave the longest walk of 64 lanes — which a deeper tree punishes at once: the Cornell box with its boxes
        // tessellated 2 x 2 per face (108 triangles, 3 levels) renders in 204 ms fused against 125 ms in the wavefront organisation,
        // whose trace kernel refills finished lanes (36 triangles, 2 levels: 58.7 against 97.5 ms; profiles/r05_size_sweep.txt)
        const bool shallow = s->dev.wide_levels <= 2u;
        mode = (fits && shallow) ? MTR_MODE_FUSED : MTR_MODE_WAVEFRONT;
    }
    *mode_io = mode;
    if (developed_ok) {          // MTR_FLAG_DEVELOPED_ROWS: the fused kernel's row flush, rows in LDS, time bins (not a phasor film)
        *developed_ok = 0u;
        if (mode == MTR_MODE_FUSED && !f.n_freq) {
            FusedArgs probe{}; FusedConfig pc{};
            probe.sc = s->dev; probe.cam = s->cam; probe.film = f; probe.rc = make_render_const(*p, f, s->dev.n_ems); probe.nlos_on = s->nlos.on ? 1u : 0u;
            if (fused_plan(s->dev, f, n_pixels, spp_chunk, usable_cus(c, p), probe, pc) && pc.hist_lds) *developed_ok = 1u;
        }
    }
    return MTR_OK;
}

extern "C" {

int mtr_film_clear(mtr_ctx *c, const mtr_film_desc *f, float *t4, float *s4)
{
    if (!c || !f) return fail(c, MTR_ERR_INVALID, "mtr_film_clear: NULL argument");
    HIP_TRY(c, hipSetDevice(c->device));
    size_t npix = (size_t)f->width * f->height;
    const Film fm = film_from_desc(*f);
    const size_t per_pixel = fm.n_freq ? (size_t)2 * fm.n_freq + 1 : (size_t)fm.bins * 4;   // (2F+1) | [lasers][T][4]
    if (t4) HIP_TRY(c, hipMemsetAsync(t4, 0, npix * per_pixel * sizeof(float), c->stream));
    if (s4) HIP_TRY(c, hipMemsetAsync(s4, 0, npix * 4 * sizeof(float), c->stream));
    return MTR_OK;
}

int mtr_render(mtr_scene *s, const mtr_render_params *p, float *t4, float *s4,
               mtr_counters *counters_out, mtr_kernel_times *times_out)
{
    if (!s || !p || !t4 || !s4) return fail(s ? s->ctx : nullptr, MTR_ERR_INVALID, "mtr_render: NULL argument");
    mtr_ctx *c = s->ctx;
    const Film &f = s->film;
    const uint64_t npix_crop = (uint64_t)f.crop_w * f.crop_h;
    if (p->spp_total == 0 || p->spp_begin > p->spp_end || p->spp_end > p->spp_total)
        return fail(c, MTR_ERR_INVALID, "mtr_render: bad sample range");
    if (p->pixel_begin > p->pixel_end || p->pixel_end > npix_crop)
        return fail(c, MTR_ERR_INVALID, "mtr_render: bad pixel range");
    if (npix_crop * p->spp_total > (1ull << 32))
        return fail(c, MTR_ERR_UNSUPPORTED, "mtr_render: W*H*spp exceeds 2^32 lanes (common.py:51); shard the render");
    if (p->max_depth < -1 || p->rr_depth <= 0) return fail(c, MTR_ERR_INVALID, "mtr_render: bad max_depth / rr_depth");
    if (p->mode > MTR_MODE_WAVEFRONT) return fail(c, MTR_ERR_INVALID, "mtr_render: unknown mode");
    HIP_TRY(c, hipSetDevice(c->device));

    FusedArgs a{};
    a.sc = s->dev; a.cam = s->cam; a.film = f;
    a.rc = make_render_const(*p, f, s->dev.n_ems);
    a.pixel_begin = p->pixel_begin; a.pixel_end = p->pixel_end;
    a.spp_begin = p->spp_begin; a.spp_chunk = p->spp_end - p->spp_begin;
    a.film_out = t4; a.steady_out = s4;
    a.counters = c->d_counters;
    a.log = s->log;
    a.nlos_on = s->nlos.on ? 1u : 0u;
    if (s->nlos.on) {
        a.nlos = s->nlos.k;
        if (s->nlos.k.film_w != f.width || s->nlos.k.film_h != f.height)
            return fail(c, MTR_ERR_INVALID, "mtr_render: film size changed after mtr_scene_set_nlos; call it again");
    }

    const uint32_t n_pixels = p->pixel_end - p->pixel_begin;
    const bool want_stats = counters_out || times_out;
    if (!(p->flags & MTR_FLAG_KEEP_COUNTERS)) HIP_TRY(c, hipMemsetAsync(c->d_counters, 0, sizeof(DevCounters), c->stream));
    if (s->log.count) HIP_TRY(c, hipMemsetAsync(s->log.count, 0, sizeof(unsigned long long), c->stream));
    if (times_out) HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
    uint32_t launches = 0, scatter_launches = 0, wf_trace_n = 0;
    float scatter_ms = 0.0f, wf_trace_ms = 0.0f, wf_shade_ms = 0.0f;
    if (n_pixels && a.spp_chunk) {
        uint32_t mode = p->mode, dev_ok = 0u;
        if (int r = resolve_mode(s, p, n_pixels, a.spp_chunk, &mode, &dev_ok)) return r;
        if ((p->flags & MTR_FLAG_DEVELOPED_ROWS) && !dev_ok)
            return fail(c, MTR_ERR_UNSUPPORTED, "mtr_render: MTR_FLAG_DEVELOPED_ROWS needs the fused organisation with time-bin rows in LDS (see mtr_render_plan)");
        if (p->n_bands && (mode == MTR_MODE_WAVEFRONT || !p->band_done || p->n_bands > n_pixels))
            return fail(c, MTR_ERR_UNSUPPORTED, "mtr_render: band completion words need the fused organisation, a band_done array and at most one band per pixel");
        if (mode == MTR_MODE_WAVEFRONT) {
            int r = wf_render(s, p, t4, s4, a.rc, &wf_trace_ms, &scatter_ms, &launches, &scatter_launches, times_out != nullptr, &wf_trace_n, want_stats, &wf_shade_ms);
            if (r) return r;
        } else {
            FusedConfig cfg{};
            if (!fused_plan(s->dev, f, n_pixels, a.spp_chunk, usable_cus(c, p), a, cfg))
                return fail(c, MTR_ERR_UNSUPPORTED, "mtr_render: no kernel configuration fits (BVH depth / LDS)");
            a.ticket = c->d_ticket + (c->fused_launches++ & 15u);
            a.n_bands = 0u;
            if (p->n_bands) {
                // (one banded launch in flight per context: the counts are the context's; callers that overlap launches on two
                // streams — the per-band pipeline — do not use band words)
                if (c->band_cap < p->n_bands) {
                    HIP_TRY(c, hipStreamSynchronize(c->stream));
                    if (c->d_band_count) (void)hipFree(c->d_band_count);
                    c->d_band_count = nullptr; c->band_cap = 0;
                    HIP_TRY(c, hipMalloc((void **)&c->d_band_count, (size_t)p->n_bands * sizeof(uint32_t)));
                    c->band_cap = p->n_bands;
                }
                HIP_TRY(c, hipMemsetAsync(c->d_band_count, 0, (size_t)p->n_bands * sizeof(uint32_t), c->stream));
                a.n_bands = p->n_bands; a.band_px = n_pixels / p->n_bands; a.band_epoch = p->band_epoch;      // (>= 1: n_bands <= n_pixels, checked above; the last band takes the remainder)
                a.band_count = c->d_band_count; a.band_done = (uint32_t *)(uintptr_t)p->band_done;
            }
            HIP_TRY(c, launch_fused(a, cfg, c->stream));
            launches = 1;
        }
    }
    if (times_out) HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
    if (want_stats) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (counters_out) {
            DevCounters h;
            HIP_TRY(c, hipMemcpy(&h, c->d_counters, sizeof h, hipMemcpyDeviceToHost));
            memset(counters_out, 0, sizeof *counters_out);
            counters_out->paths = h.paths; counters_out->rays_closest = h.rays_closest;
            counters_out->rays_shadow = h.rays_shadow; counters_out->splats_issued = h.splats_issued;
            counters_out->bounces = h.bounces; counters_out->splats_overflow = h.splats_overflow;
            counters_out->reserved[0] = h.r0; counters_out->reserved[1] = h.r1;
        }
        if (times_out) {
            memset(times_out, 0, sizeof *times_out);
            float ms = 0.0f;
            HIP_TRY(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
            times_out->total_ms = ms; times_out->trace_ms = ms - scatter_ms; times_out->scatter_ms = scatter_ms;
            times_out->trace_launches = launches; times_out->scatter_launches = scatter_launches;
            times_out->wf_trace_ms = wf_trace_ms; times_out->wf_trace_kernel_launches = wf_trace_n; times_out->wf_shade_ms = wf_shade_ms;
        }
    }
    return MTR_OK;
}

int mtr_render_plan(mtr_scene *s, const mtr_render_params *p, uint32_t *mode_out, uint32_t *developed_rows_ok)
{
    if (!s || !p || !mode_out) return fail(s ? s->ctx : nullptr, MTR_ERR_INVALID, "mtr_render_plan: NULL argument");
    if (p->mode > MTR_MODE_WAVEFRONT) return fail(s->ctx, MTR_ERR_INVALID, "mtr_render_plan: unknown mode");
    if (p->pixel_begin > p->pixel_end || p->spp_begin > p->spp_end) return fail(s->ctx, MTR_ERR_INVALID, "mtr_render_plan: bad range");
    uint32_t mode = p->mode;
    if (int r = resolve_mode(s, p, p->pixel_end - p->pixel_begin, p->spp_end - p->spp_begin, &mode, developed_rows_ok)) return r;
    *mode_out = mode;
    return MTR_OK;
}

int mtr_counters_reset(mtr_ctx *c)
{
    if (!c) return fail(c, MTR_ERR_INVALID, "mtr_counters_reset: NULL argument");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemsetAsync(c->d_counters, 0, sizeof(DevCounters), c->stream));
    return MTR_OK;
}

int mtr_counters_read(mtr_ctx *c, mtr_counters *out)
{
    if (!c || !out) return fail(c, MTR_ERR_INVALID, "mtr_counters_read: NULL argument");
    HIP_TRY(c, hipSetDevice(c->device));
    DevCounters h;
    HIP_TRY(c, hipMemcpy(&h, c->d_counters, sizeof h, hipMemcpyDeviceToHost));
    memset(out, 0, sizeof *out);
    out->paths = h.paths; out->rays_closest = h.rays_closest; out->rays_shadow = h.rays_shadow;
    out->splats_issued = h.splats_issued; out->bounces = h.bounces; out->splats_overflow = h.splats_overflow;
    out->reserved[0] = h.r0; out->reserved[1] = h.r1;
    return MTR_OK;
}

int mtr_film_develop(mtr_ctx *c, const mtr_film_desc *fd, const float *t4, float *t3, const float *s4, float *s3)
{
    if (!c || !fd) return fail(c, MTR_ERR_INVALID, "mtr_film_develop: NULL argument");
    int rc = check_film(c, *fd);
    if (rc) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, launch_develop(film_from_desc(*fd), t4, t3, s4, s3, c->stream));
    return MTR_OK;
}

int mtr_splat_add(mtr_ctx *c, const mtr_splat_soa *s, const mtr_film_desc *fd, int variant, float *t4, float *elapsed_ms)
{
    if (!c || !s || !fd || !t4) return fail(c, MTR_ERR_INVALID, "mtr_splat_add: NULL argument");
    int rc = check_film(c, *fd);
    if (rc) return rc;
    const bool film_zero = (variant & MTR_SPLAT_FILM_ZERO) != 0;
    variant &= ~MTR_SPLAT_FILM_ZERO;
    if (variant != 0 && variant != 1) return fail(c, MTR_ERR_INVALID, "mtr_splat_add: variant must be 0 or 1 (| MTR_SPLAT_FILM_ZERO)");
    if (s->n && (!s->pixel || !s->opl || !s->r || !s->g || !s->b))
        return fail(c, MTR_ERR_INVALID, "mtr_splat_add: NULL splat array");
    HIP_TRY(c, hipSetDevice(c->device));
    Film fm = film_from_desc(*fd);
    if (fm.n_freq) {
        if (c->freq_cap < fm.n_freq) {
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            if (c->d_freq) (void)hipFree(c->d_freq);
            c->d_freq = nullptr; c->freq_cap = 0;
            HIP_TRY(c, hipMalloc((void **)&c->d_freq, (size_t)fm.n_freq * 4));
            c->freq_cap = fm.n_freq;
        }
        HIP_TRY(c, hipMemcpyAsync(c->d_freq, fd->frequencies, (size_t)fm.n_freq * 4, hipMemcpyHostToDevice, c->stream));
        fm.freq = c->d_freq;
    }
    void *scratch = nullptr;
    if (variant == 1 && !fm.n_freq) {
        const size_t need = 8u * ((size_t)fm.width * fm.height + 2u);
        if (c->runs_cap < need) {
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            if (c->d_runs) (void)hipFree(c->d_runs);
            c->d_runs = nullptr; c->runs_cap = 0;
            HIP_TRY(c, hipMalloc(&c->d_runs, need));
            c->runs_cap = need;
        }
        scratch = c->d_runs;
    }
    if (elapsed_ms) HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
    // variant 1: pixel-sorted input goes through the LDS rows as it is.  Anything else is PARTITIONED BY PIXEL on the device first
    // (mtr_splat.hip: two scatter passes over 16-byte records, then the same rows) when the film's shape allows it; that path
    // reads the sortedness flag back — one stream synchronisation — and holds 32 bytes of workspace per contribution for the call
    const bool can_partition = variant == 1 && scratch && splat_partition_supported(*s, fm);
    HIP_TRY(c, launch_splat_add(variant, *s, fm, t4, nullptr, scratch, c->stream, film_zero && variant == 1, !can_partition));
    bool second_leg = false;
    if (can_partition && s->n) {
        uint32_t unsorted = 0;
        if (elapsed_ms) HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
        HIP_TRY(c, hipMemcpyAsync(&unsorted, scratch, 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (unsorted) {
            // The workspace (32 B per contribution: 32 GiB for 2^30 of them) is raw device memory that torch's caching allocator
            // cannot see.  At most kPartRetain of it stays with the context between calls; a larger one is released before the
            // call returns (one more stream synchronisation — the call has blocked once already to read the flag).
            constexpr size_t kPartRetain = (size_t)256 << 20;
            const size_t need = splat_partition_scratch_bytes(*s, fm);
            hipError_t e = hipSuccess;
            if (c->part_cap < need) {
                if (c->d_part) (void)hipFree(c->d_part);
                c->d_part = nullptr; c->part_cap = 0;
                e = hipMalloc(&c->d_part, need);
                if (e == hipSuccess) c->part_cap = need; else { c->d_part = nullptr; (void)hipGetLastError(); }
            }
            // (timed calls: the passes below get their own pair of events — the allocation above is host time, not kernel time)
            if (elapsed_ms) { HIP_TRY(c, hipEventRecord(c->ev2, c->stream)); second_leg = true; }
            if (e != hipSuccess) HIP_TRY(c, launch_splat_add(0, *s, fm, t4, nullptr, nullptr, c->stream));     // no room for the workspace: the contract form
            else HIP_TRY(c, launch_splat_partitioned(*s, fm, t4, film_zero, nullptr, c->d_part, c->n_cu, c->stream));
            if (elapsed_ms) HIP_TRY(c, hipEventRecord(c->ev3, c->stream));
            if (c->part_cap > kPartRetain) {
                HIP_TRY(c, hipStreamSynchronize(c->stream));
                (void)hipFree(c->d_part);
                c->d_part = nullptr; c->part_cap = 0;
            }
        }
    } else if (elapsed_ms) HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
    if (elapsed_ms) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        HIP_TRY(c, hipEventElapsedTime(elapsed_ms, c->ev0, c->ev1));
        if (second_leg) {
            float ms2 = 0.0f;
            HIP_TRY(c, hipEventElapsedTime(&ms2, c->ev2, c->ev3));
            *elapsed_ms += ms2;
        }
    }
    return MTR_OK;
}

int mtr_ctx_trim(mtr_ctx *c)
{
    if (!c) return MTR_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->d_part) (void)hipFree(c->d_part);
    c->d_part = nullptr; c->part_cap = 0;
    return MTR_OK;
}

int mtr_debug_set_splat_log(mtr_scene *s, uint32_t *log_device, uint64_t capacity, uint64_t *n_records_device)
{
    if (!s) return MTR_ERR_INVALID;
    s->log.rec = log_device; s->log.cap = capacity; s->log.count = (unsigned long long *)n_records_device;
    if (!log_device || !n_records_device) { s->log.rec = nullptr; s->log.cap = 0; s->log.count = nullptr; }
    return MTR_OK;
}

} // extern "C"
