// mtr_api.hip — the C-ABI of include/mitransient_amd.h.
//
// Host side of the library: scene ingestion (derived per-triangle frames, emitter normals,
// BVH2 build, upload), render planning and launches, film develop/clear, the stand-alone
// scatter-add.  There is no CPU execution path: without a HIP device every entry point fails.
#include "../../include/mitransient_amd.h"
#include "mtr_scene_host.h"
#include "mtr_core.h"
#include "mtr_kernels.h"

#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

using namespace mtr;

struct mtr_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    int n_cu = 256;
    std::string err;
    DevCounters *d_counters = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

struct mtr_scene {
    mtr_ctx *ctx = nullptr;
    SceneDev dev{};
    Camera cam{};
    Film film{};
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
    HIP_TRY(nullptr, hipEventCreate(&c->ev0));
    HIP_TRY(nullptr, hipEventCreate(&c->ev1));
    *out = c;
    return MTR_OK;
}

void mtr_ctx_destroy(mtr_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->d_counters) (void)hipFree(c->d_counters);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
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
    s->film = hs.film; s->cam = hs.cam;

#define UP(vec, field)                                                       \
    do { rc = upload(s, vec, &s->dev.field); if (rc) { mtr_scene_destroy(s); return rc; } } while (0)
    UP(hs.nodes, nodes); UP(hs.tgeom, tgeom); UP(hs.tshade, tshade); UP(hs.mats, mats); UP(hs.ems, ems);
#undef UP
    s->dev.n_nodes = (uint32_t)hs.nodes.size(); s->dev.n_tris = d->n_tris;
    s->dev.n_mats = d->n_materials; s->dev.n_ems = d->n_emitters;
    s->dev.bvh_depth = hs.bvh_depth; s->n_leaves = hs.n_leaves;
    *out = s;
    return MTR_OK;
}

void mtr_scene_destroy(mtr_scene *s)
{
    if (!s) return;
    if (s->ctx) (void)hipSetDevice(s->ctx->device);
    for (void *p : s->allocs) (void)hipFree(p);
    delete s;
}

int mtr_scene_set_film(mtr_scene *s, const mtr_film_desc *f)
{
    if (!s || !f) return fail(s ? s->ctx : nullptr, MTR_ERR_INVALID, "mtr_scene_set_film: NULL argument");
    int rc = check_film(s->ctx, *f);
    if (rc) return rc;
    s->film = film_from_desc(*f);
    return MTR_OK;
}

int mtr_scene_bvh_info(const mtr_scene *s, uint32_t *n_nodes, uint32_t *max_depth, uint32_t *n_leaves)
{
    if (!s) return MTR_ERR_INVALID;
    if (n_nodes) *n_nodes = s->dev.n_nodes;
    if (max_depth) *max_depth = s->dev.bvh_depth;
    if (n_leaves) *n_leaves = s->n_leaves;
    return MTR_OK;
}

int mtr_film_clear(mtr_ctx *c, const mtr_film_desc *f, float *t4, float *s4)
{
    if (!c || !f) return fail(c, MTR_ERR_INVALID, "mtr_film_clear: NULL argument");
    HIP_TRY(c, hipSetDevice(c->device));
    size_t npix = (size_t)f->width * f->height;
    if (t4) HIP_TRY(c, hipMemsetAsync(t4, 0, npix * f->temporal_bins * 4 * sizeof(float), c->stream));
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

    const uint32_t n_pixels = p->pixel_end - p->pixel_begin;
    const bool want_stats = counters_out || times_out;
    HIP_TRY(c, hipMemsetAsync(c->d_counters, 0, sizeof(DevCounters), c->stream));
    if (s->log.count) HIP_TRY(c, hipMemsetAsync(s->log.count, 0, sizeof(unsigned long long), c->stream));
    if (times_out) HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
    uint32_t launches = 0;
    if (n_pixels && a.spp_chunk) {
        if (p->mode == MTR_MODE_WAVEFRONT)
            return fail(c, MTR_ERR_UNSUPPORTED, "mtr_render: MTR_MODE_WAVEFRONT is not built in this revision");
        FusedConfig cfg{};
        if (!fused_plan(s->dev, f, n_pixels, a.spp_chunk, c->n_cu, a, cfg))
            return fail(c, MTR_ERR_UNSUPPORTED, "mtr_render: no kernel configuration fits (BVH depth / LDS)");
        HIP_TRY(c, launch_fused(a, cfg, c->stream));
        launches = 1;
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
            times_out->total_ms = ms; times_out->trace_ms = ms; times_out->trace_launches = launches;
        }
    }
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
    if (variant != 0 && variant != 1) return fail(c, MTR_ERR_INVALID, "mtr_splat_add: variant must be 0 or 1");
    if (s->n && (!s->pixel || !s->opl || !s->r || !s->g || !s->b))
        return fail(c, MTR_ERR_INVALID, "mtr_splat_add: NULL splat array");
    HIP_TRY(c, hipSetDevice(c->device));
    if (elapsed_ms) HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
    HIP_TRY(c, launch_splat_add(variant, *s, film_from_desc(*fd), t4, nullptr, c->stream));
    if (elapsed_ms) {
        HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        HIP_TRY(c, hipEventElapsedTime(elapsed_ms, c->ev0, c->ev1));
    }
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
