// mtr_kernels.h — host-visible launch interface of the HIP kernels (internal to the library).
#pragma once
#include <hip/hip_runtime.h>
#include "mtr_core.h"

namespace mtr {

constexpr int kBlock = 256;          // 4 wave64 per workgroup

struct DevCounters {                 // device mirror of mtr_counters (u64 atomics)
    unsigned long long paths, rays_closest, rays_shadow, splats_issued, bounces, splats_overflow, r0, r1;
};

struct SceneDev {
    const Node *nodes; uint32_t n_nodes;
    const TriGeom *tgeom; const TriShade *tshade; uint32_t n_tris;
    const mtr_material *mats; uint32_t n_mats;
    const Emitter *ems; uint32_t n_ems;
    uint32_t bvh_depth;
    uint32_t lds_bytes;              // bytes needed to stage the whole scene in LDS
};

struct SplatLog { uint32_t *rec; unsigned long long cap; unsigned long long *count; };

struct FusedArgs {
    SceneDev sc;
    Camera cam;
    Film film;
    RenderConst rc;
    uint32_t pixel_begin, pixel_end;     // crop-window pixels
    uint32_t spp_begin, spp_chunk;       // samples [spp_begin, spp_begin + spp_chunk)
    uint32_t G;                          // pixels per segment (one workgroup owns a segment)
    uint32_t nseg;
    float *film_out;                     // (H, W, T, 4)
    float *steady_out;                   // (H, W, 4)
    DevCounters *counters;
    SplatLog log;
};

struct FusedConfig { int stack; bool scene_lds; bool hist_lds; size_t lds_bytes; int grid; };

// chooses G, LDS carve-up and grid for a render; returns false if nothing fits
bool fused_plan(const SceneDev &sc, const Film &film, uint32_t n_pixels, uint32_t spp_chunk, int n_cu,
                FusedArgs &args, FusedConfig &cfg);
hipError_t launch_fused(const FusedArgs &args, const FusedConfig &cfg, hipStream_t stream);

hipError_t launch_splat_add(int variant, const mtr_splat_soa &s, const Film &film, float *film_out,
                            DevCounters *counters, hipStream_t stream);
hipError_t launch_develop(const Film &film, const float *t4, float *t3, const float *s4, float *s3, hipStream_t stream);

} // namespace mtr
