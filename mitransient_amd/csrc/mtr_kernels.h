// mtr_kernels.h — host-visible launch interface of the HIP kernels (internal to the library).
#pragma once
#include <hip/hip_runtime.h>
#include "mtr_core.h"
#include "mtr_nlos.h"

namespace mtr {

#ifndef MTR_BLOCK
#define MTR_BLOCK 256
#endif
constexpr int kBlock = MTR_BLOCK;    // 4 wave64 per workgroup (experiments: -DMTR_BLOCK=320 with -DMTR_FUSED_MIN_WAVES=5)

struct DevCounters {                 // device mirror of mtr_counters (u64 atomics)
    unsigned long long paths, rays_closest, rays_shadow, splats_issued, bounces, splats_overflow, r0, r1;
};

struct SceneDev {
    const Node *nodes; uint32_t n_nodes;
    const WNode *wnodes; uint32_t n_wnodes;                            // 8-wide tree over the same leaves (null for large scenes)
    const QNode4 *wnodes4; uint32_t n_wnodes4;                         // quantised 4-wide tree (always present)
    const QNode8 *wnodes8q; uint32_t n_wnodes8q;                       // quantised 8-wide tree (scenes walked in HBM; null: walk wnodes4)
    const TriPair *tpairs; const TriShade *tshade; uint32_t n_slots;   // triangle slots (even), see mtr_core.h
    const mtr_material *mats; uint32_t n_mats;
    const Emitter *ems; uint32_t n_ems;
    const q4 *samp_tris; const float *face_pmf, *face_cdf;   // mesh-emitter sampling tables (HBM; null without mesh emitters)
    const q4 *samp_vn;               // ... vertex normals of those meshes (null unless a mesh emitter has them)
    const q4 *vnormals;              // [3 * n_slots] vertex normals of smooth-shaded slots (HBM; null when every triangle is flat)
    const q4 *texels, *tex_info, *uvs;   // bitmap textures (HBM; null without): see SceneView
    uint32_t bvh_depth;
    uint32_t wide_levels, wide4_levels, wide8q_levels;   // levels of wnodes / wnodes4 / wnodes8q: a wide walk stacks at most one group per level
    uint32_t lds_bytes;              // bytes needed to stage the whole scene in LDS
    uint32_t traits;                 // kTr* bits (mtr_core.h) that hold for the material / emitter tables: kernels specialised on them are chosen
    uint32_t has_rough;              // the scene needs the EXTENDED shading code (kernels instantiated with ROUGH = true): a material
                                     // is a GGX lobe (MTR_BSDF_ROUGH*), or a triangle is smooth-shaded (vnormals)
    FlatTop flat;                    // traits & kTrFlatTop: the top level as scalar kernel arguments (mtr_core.h)
};

// streaming accesses (data touched once): non-temporal 16-byte load / store
__device__ __forceinline__ uint4 nt_load(const uint4 *p)
{
    const uint32_t *w = (const uint32_t *)p;
    return make_uint4(__builtin_nontemporal_load(w), __builtin_nontemporal_load(w + 1), __builtin_nontemporal_load(w + 2), __builtin_nontemporal_load(w + 3));
}
__device__ __forceinline__ float4 nt_load(const float4 *p)
{
    const float *w = (const float *)p;
    return make_float4(__builtin_nontemporal_load(w), __builtin_nontemporal_load(w + 1), __builtin_nontemporal_load(w + 2), __builtin_nontemporal_load(w + 3));
}
__device__ __forceinline__ void nt_store(float4 *p, float4 v)
{
    float *w = (float *)p;
    __builtin_nontemporal_store(v.x, w); __builtin_nontemporal_store(v.y, w + 1); __builtin_nontemporal_store(v.z, w + 2); __builtin_nontemporal_store(v.w, w + 3);
}

struct SplatLog { uint32_t *rec; unsigned long long cap; unsigned long long *count; };

struct FusedArgs {
    SceneDev sc;
    Camera cam;
    Film film;
    RenderConst rc;
    uint32_t pixel_begin, pixel_end;     // crop-window pixels
    uint32_t spp_begin, spp_chunk;       // samples [spp_begin, spp_begin + spp_chunk)
    uint32_t G;                          // row slots of a workgroup's pixel ring (k_fused)
    FastDiv div_spp, div_G;              // i / spp_chunk, q / G without the ~35-instruction integer division
    uint32_t stack_rows;                 // rows of the per-lane LDS stack columns: levels of the wide tree walked + 1 (>= 3)
    uint32_t chunk, n_chunks;            // consecutive pixels per work ticket, number of tickets of this launch
    uint32_t *ticket;                    // device counter, zeroed before the launch (launch_fused)
    float *film_out;                     // (H, W, T, 4)
    float *steady_out;                   // (H, W, 4)
    DevCounters *counters;
    SplatLog log;
    uint32_t nlos_on;                    // 1: transient_nlos_path + nlos_capture_meter (nlos valid)
    float fixed_lim; uint32_t fixed_dcap;    // MTR_FLAG_DETERMINISTIC rows: what is summed in fixed point (LdsFixedSink's RANGE GUARD; fused_plan)
    // band completion words (mtr_render_params.n_bands): band b = pixels [b * band_px, (b + 1) * band_px) of the launch's range;
    // band_count[b] counts its flushed pixels (zeroed before the launch), band_done[b] receives band_epoch when it is complete
    uint32_t n_bands, band_px, band_epoch;
    uint32_t *band_count, *band_done;
    NlosConst nlos;
};

struct FusedConfig { int stack; bool scene_lds; bool hist_lds; bool fixed; bool rough; size_t lds_bytes; int grid; int per_cu; uint32_t traits; };

// chooses G, LDS carve-up and grid for a render; returns false if nothing fits
bool fused_plan(const SceneDev &sc, const Film &film, uint32_t n_pixels, uint32_t spp_chunk, int n_cu,
                FusedArgs &args, FusedConfig &cfg);
hipError_t launch_fused(const FusedArgs &args, const FusedConfig &cfg, hipStream_t stream);
// NLOS prepare (transientnlospath.py:295-336): one ray per film pixel -> scanned points, + the laser's axis
hipError_t launch_nlos_prepare(const SceneDev &sc, const NlosConst &nc, q4 *targets, hipStream_t stream);

// ---- MTR_MODE_WAVEFRONT (mtr_wavefront.hip) ---------------------------------------------------
constexpr uint32_t kWfKeys = 5;        // material-type queues: diffuse, conductor, dielectric, none, miss

struct WfArgs {
    SceneDev sc;
    Camera cam;
    Film film;
    RenderConst rc;
    uint32_t pix0, P;                    // tile: crop-window pixels [pix0, pix0 + P)
    uint32_t spp_begin, S;               // samples [spp_begin, spp_begin + S) of every pixel of the tile
    uint32_t n_slots;                    // P * S
    uint32_t G, seg, n_seg;              // segment = G whole pixels = G * S slots, owned by one workgroup per launch
    uint32_t parity;                     // which of the two live lists this bounce reads
    uint32_t *ticket; uint32_t ticket_cur;   // segment tickets: two counters alternating between consecutive launches
    float *planes;                       // SoA state, PL_COUNT planes of n_slots
    uint32_t *q_live;                    // [2][n_slots] ping-pong live lists (slot indices), segment sg at sg * seg
    float4 *q_ray;                       // [2][n_slots][2] the rays of the live lists IN LIST ORDER: (o, tmax) (d, eta)
    uint32_t *seg_live;                  // [2][n_seg]   their lengths
    uint32_t *seg_list;                  // [2][n_seg]   the segments that HOLD live paths, per parity (the kernels of a bounce walk this list:
    uint32_t *seg_list_n;                // [2]          its length        a deep bounce of max_depth 65 touches a handful of 32768 segments)
    uint32_t *q_mat;                     // [kWfKeys][n_slots] material-sorted hit lists, same segmentation
    uint32_t *q_shadow;                  // (unused since round 6: the occlusion results go to the rays' list positions, no slot needed)
    float4 *r_shadow;                    // [n_slots][2] those shadow rays, in list order: (o, tmax) (d, -)
    uint32_t *seg_shadow;                // [n_seg] their counts
    uint32_t *q_zombie;                  // [2][n_slots] scenes in HBM: paths that ended with an emitter-sampling term parked (Q_PEND), per parity
    uint32_t *seg_zombie;                // [2][n_seg] their counts
    uint8_t *occ;                        // [n_seg][occ_stride] shadow-ray results in the order of the segment's shadow list: 1 = occluded
    uint32_t occ_stride;                 // seg rounded up to 16 (the flags of a segment leave LDS as 16-byte stores)
    // TRACE ORDER (round 6; an experiment that lost and is off by default — mtr_api.hip wf_render): k_wf_shade sorts the positions of a segment's next live list by (cell of the ray's origin, octant of its
    // direction) — a counting sort in LDS — and k_wf_trace fetches its rays in that order: the 64 rays of a wave start in one part
    // of the scene and walk it the same way round.  The list itself, and everything keyed by the list position, stays in arrival order.
    uint16_t *q_order;                   // [n_slots] j-th ray to trace -> its position in the live list (per segment); null: list order
    uint16_t *q_order_sh;                // [n_slots] the same for the shadow list
    float sort_lo[3], sort_scale[3];     // origin -> cell: floor((o - lo) * scale), clamped to the grid
    uint32_t sort_bits[3];               // log2 of the grid's extent per axis (5 bits in all)
    uint32_t trace_any;                  // k_wf_trace: 0 closest hits of the live lists, 1 occlusion of the shadow lists
    uint32_t first_bounce;               // k_wf_shade: this launch shades bounce 0 — the path state is rebuilt from (pixel, sample), not loaded
    uint32_t *seg_mat;                   // [n_seg][kWfKeys] their lengths
    uint32_t *live_total;                // optional: number of survivors of this bounce (unbounded-depth renders)
    uint4 *rec;                          // [P][rec_cap] time-bin records (bin, r, g, b)
    uint32_t *rec_count;                 // [P]
    uint32_t rec_cap;                    // 0: rows do not fit LDS -> contributions go straight to HBM atomics
    uint32_t film_zero;                  // film rows of this tile are zero on entry: the row flush stores
    float *film_out, *steady_out;
    DevCounters *counters;
    SplatLog log;
    uint32_t nlos_on;                    // NLOS tier in the wavefront organisation: raygen = nlos_begin, one k_wf_nlos_bounce per bounce
    NlosConst nlos;
};
struct WfConfig { int stack; bool scene_lds; size_t lds_bytes; };

size_t wf_planes_bytes(uint32_t n_slots);
bool wf_plan(const SceneDev &sc, WfConfig &cfg);
// which: 0 raygen, 1 trace (closest hit + material-sorted queues, or occlusion when a.trace_any), 2 shade,
// 3 time-bin scatter-add, 5 one whole NLOS bounce (a.nlos_on)
hipError_t launch_wf(const WfArgs &a, const WfConfig &cfg, int which, int grid, hipStream_t stream);

// scratch (variant 1): device buffer of >= 8 * (width * height + 2) bytes for the run table; NULL forces the atomics
hipError_t launch_splat_add(int variant, const mtr_splat_soa &s, const Film &film, float *film_out,
                            DevCounters *counters, void *scratch, hipStream_t stream, bool film_zero = false, bool fallback_atomics = true);
// device-side partition by pixel in front of the row kernel (mtr_splat.hip)
bool splat_partition_supported(const mtr_splat_soa &s, const Film &film);
size_t splat_partition_scratch_bytes(const mtr_splat_soa &s, const Film &film);
hipError_t launch_splat_partitioned(const mtr_splat_soa &s, const Film &film, float *film_out, bool film_zero, DevCounters *counters,
                                    void *scratch, int n_cu, hipStream_t stream);
hipError_t launch_develop(const Film &film, const float *t4, float *t3, const float *s4, float *s3, hipStream_t stream);

} // namespace mtr
