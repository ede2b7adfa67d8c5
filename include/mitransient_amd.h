/*
 * mitransient_amd.h — C-ABI of the MI355X-native transient path tracer.
 *
 * This is the drop-in boundary for ONE path of diegoroyo/mitransient: the
 * `transient_path` integrator + `transient_hdr_film` time-binning
 * (reference: mitransient/integrators/common.py:122-213,
 *  mitransient/integrators/transientpath.py:88-326,
 *  mitransient/films/transient_hdr_film.py:210-276,
 *  mitransient/render/transient_image_block.py:56-151).
 *
 * The reference has no FFI of its own: its Python plugins call into Mitsuba 3
 * (C++) / Dr.Jit through pybind11 objects.  The entry points below are what a
 * ctypes binding placed at those Python call sites would bind (see
 * INTEGRATION.md for the stub); each one cites the reference interface it
 * replaces.
 *
 * Conventions: opaque handles; plain pointers and sizes; no C++/torch types;
 * every function returns 0 on success or a negative mtr_status; the message of
 * the last failure on a context is available through mtr_last_error().
 * Pointers documented "device" must be HIP device pointers on the context's
 * GPU; all launches go to the hipStream_t handed to mtr_ctx_set_stream()
 * (default: the NULL stream).  A context is not re-entrant; use one per
 * thread/stream.
 */
#ifndef MITRANSIENT_AMD_H
#define MITRANSIENT_AMD_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MTR_ABI_VERSION 12

typedef enum mtr_status {
    MTR_OK = 0,
    MTR_ERR_INVALID = -1,     /* bad argument / unsupported property            */
    MTR_ERR_NO_DEVICE = -2,   /* no HIP device: there is NO CPU fallback        */
    MTR_ERR_HIP = -3,         /* a HIP runtime call failed                      */
    MTR_ERR_OOM = -4,
    MTR_ERR_UNSUPPORTED = -5
} mtr_status;

/* ---- materials: BSDF subset of the north-star path --------------------- */
enum { MTR_BSDF_DIFFUSE = 0, MTR_BSDF_CONDUCTOR = 1, MTR_BSDF_DIELECTRIC = 2,
       MTR_BSDF_NULL = 3 /* no BSDF: absorbs */,
       MTR_BSDF_ROUGHCONDUCTOR = 4, /* microfacet conductor, isotropic alpha, visible-normal sampling (mitsuba `roughconductor`,
                                       distribution = ggx | beckmann (MTR_MAT_BECKMANN), sample_visible = true): a smooth lobe,
                                       takes part in emitter sampling */
       MTR_BSDF_ROUGHPLASTIC = 5,   /* microfacet dielectric coat over a diffuse base (mitsuba `roughplastic`) */
       MTR_BSDF_ROUGHDIELECTRIC = 6, /* rough refractive interface (mitsuba `roughdielectric`, ABI 11): int_ior / ext_ior, c = specular
                                       reflectance, c2 = specular transmittance, alpha (MTR_MAT_ANISOTROPIC: alpha_v in b[0]);
                                       transmissive, never under MTR_MAT_TWOSIDED */
       MTR_BSDF_THINDIELECTRIC = 7, /* thin dielectric slab (mitsuba `thindielectric`, ABI 11): two delta lobes, no refraction, eta stays 1;
                                       int_ior / ext_ior, c / c2 = specular reflectance / transmittance */
       MTR_BSDF_PLASTIC = 8         /* smooth dielectric coat over a diffuse base (mitsuba `plastic`, ABI 11): a = diffuse_reflectance,
                                       c = specular_reflectance, int_ior / ext_ior, MTR_MAT_NONLINEAR, internal_reflectance =
                                       fresnel_diffuse_reflectance(1 / eta), specular_sampling_weight                              */ };
enum { MTR_MAT_TWOSIDED = 1u,
       MTR_MAT_NONLINEAR = 2u, /* roughplastic `nonlinear`: diffuse / (1 - diffuse * internal_reflectance) per channel */
       MTR_MAT_BECKMANN = 4u,  /* rough lobes (ABI 11): the Beckmann distribution — mitsuba's default `distribution` — instead of GGX */
       MTR_MAT_ANISOTROPIC = 8u /* roughconductor (ABI 11): `alpha` is alpha_u (along the shading frame's tangent, dp/du) and c2[0]
                                   holds alpha_v (a field conductors do not use otherwise)                                    */ };
#define MTR_ROUGH_TRANSMITTANCE_RES 64

typedef struct mtr_material {
    uint32_t type;        /* MTR_BSDF_*                                          */
    uint32_t flags;       /* MTR_MAT_*                                           */
    float    a[3];        /* diffuse: reflectance rgb | (rough)conductor: eta rgb | roughplastic: diffuse_reflectance rgb */
    float    b[3];        /* (rough)conductor: k rgb | roughdielectric with MTR_MAT_ANISOTROPIC: b[0] = alpha_v */
    float    c[3];        /* conductor/dielectric/rough*: specular_reflectance rgb */
    float    int_ior;     /* dielectric, roughplastic                            */
    float    ext_ior;     /* dielectric, roughplastic                            */
    float    c2[3];       /* dielectric: specular_transmittance rgb | roughconductor with MTR_MAT_ANISOTROPIC: c2[0] = alpha_v */
    /* rough lobes (ABI 8) */
    float    alpha;       /* roughness of the lobe (alpha_u = alpha_v unless MTR_MAT_ANISOTROPIC) */
    float    internal_reflectance;   /* roughplastic: mean rough reflectance of the coat seen from inside          */
    float    specular_sampling_weight; /* roughplastic: s_mean / (d_mean + s_mean)                                 */
    uint32_t albedo_texture;  /* 1 + index into mtr_scene_desc.textures of a bitmap that replaces `a` (diffuse reflectance,
                                 roughplastic diffuse_reflectance) at the hit's texture coordinate; 0 = the constant `a` */
    /* roughplastic: transmittance of the rough coat for cos(theta) = max(1e-6, k / 63), k = 0 .. 63 — mitsuba computes
     * this table when the plugin is built (RoughPlastic::parameters_changed: eval_transmittance by Gauss-Legendre
     * quadrature over visible normals); the caller does the same (mitransient_amd/scene.py: rough_plastic_tables) and the
     * library only interpolates it */
    float    external_transmittance[MTR_ROUGH_TRANSMITTANCE_RES];
} mtr_material;

/* `bitmap` texture (mitsuba: filter_type = bilinear, wrap_mode = repeat, to_uv = identity): linear RGB */
typedef struct mtr_texture {
    uint32_t width, height;
    const float *rgb;      /* host, height * width * 3 floats, row 0 first (v = 0), already linear (sRGB decoded by the caller) */
} mtr_texture;

/* ---- emitters: `area` emitter attached to a `rectangle` (analytic sampling) or to a triangle mesh ---- */
typedef struct mtr_emitter {
    float center[3];      /* rectangle: to_world * (0,0,0)                       */
    float du[3];          /* rectangle: to_world * (1,0,0) - center  (half edge) */
    float dv[3];          /* rectangle: to_world * (0,1,0) - center  (half edge) */
    float radiance[3];
    uint32_t is_mesh;     /* 1: the emitter is the triangle range below (obj / cube shapes) */
    uint32_t first_tri, n_tris;
    uint32_t flip_normals; /* rectangle (ABI 9): the shape's `flip_normals` — the emitting side is -normalize(du x dv);
                              sample positions are unchanged (mitsuba negates the normal, not the parameterisation).
                              A mesh emitter's flip is the winding of its triangles. */
} mtr_emitter;

/* ---- sensor: `perspective` (utils.py:92-105 of the reference) ----------- */
typedef struct mtr_camera {
    float sample_to_camera[16]; /* row-major 4x4 projective, film sample [0,1]^2 -> camera near plane */
    float to_world[16];         /* row-major 4x4 camera -> world (rigid)         */
    float near_clip, far_clip;
} mtr_camera;

/* ---- film: `transient_hdr_film` (transient_hdr_film.py:114-121) --------- */
typedef struct mtr_film_desc {
    uint32_t width, height;               /* full film size (tensor is H x W x T x 4)  */
    uint32_t crop_width, crop_height;     /* sampled window                            */
    uint32_t crop_offset_x, crop_offset_y;
    uint32_t temporal_bins;               /* T  (default 2048)                         */
    float    start_opl;                   /* default 0                                 */
    float    bin_width_opl;               /* default 0.003                             */
    /* `exhaustive_scan` (transient_hdr_film.py:119-121, transient_image_block.py:63-66,134-140): both > 0 makes
     * the tensor H x W x laser_scan_height x laser_scan_width x T x 4 with the reference's flat index
     * ((((y*W + x)*laser_scan_width + laser_x)*laser_scan_height + laser_y)*T + t)*4; 0/0 = plain H x W x T x 4 */
    uint32_t laser_scan_width, laser_scan_height;
    /* `phasor_hdr_film` (mitransient/films/phasor_hdr_film.py:125-139, render/phasor_image_block.py:42-67): n_frequencies > 0
     * makes the tensor H x W x (2F + 1) — per frequency the real and imaginary part of sum(value * exp(i * phase)),
     * phase = fmod(-2 pi f (opl - start_opl), 2 pi), then the weight channel — instead of time bins; every finite optical
     * path length counts (temporal_bins / bin_width_opl only choose the frequencies, on the host).  Monochromatic: the
     * value is channel 0 of the contribution.  frequencies: host pointer, read during the call that takes the desc. */
    uint32_t n_frequencies;
    const float *frequencies;
} mtr_film_desc;

/* ---- NLOS tier: `transient_nlos_path` + `nlos_capture_meter` + `projector` ---------------
 * (reference: mitransient/integrators/transientnlospath.py:200-249 properties, :251-383 prepare;
 *  mitransient/sensors/nloscapturemeter.py:93-202; mitsuba's `projector` emitter)            */
enum { MTR_CAPTURE_SINGLE = 1, MTR_CAPTURE_CONFOCAL = 2,
       MTR_CAPTURE_EXHAUSTIVE = 3 /* every scanned point x every illuminated point: needs an exhaustive_scan film */ };
enum { MTR_NLOS_LASER_SAMPLING = 1u,          /* nlos_laser_sampling                              */
       MTR_NLOS_HG_SAMPLING = 2u,             /* nlos_hidden_geometry_sampling                    */
       MTR_NLOS_HG_RROULETTE = 4u,            /* nlos_hidden_geometry_sampling_do_rroulette       */
       MTR_NLOS_HG_INCLUDES_WALL = 8u,        /* nlos_hidden_geometry_sampling_includes_relay_wall*/
       MTR_NLOS_ACCOUNT_FIRST_LAST = 16u,     /* account_first_and_last_bounces                   */
       MTR_NLOS_DISCARD_DIRECT = 32u,         /* discard_direct_paths                             */
       MTR_NLOS_FORCE_EQUAL_GRIDS = 64u       /* force_equal_illumination_scanning (Exhaustive)   */ };

enum { MTR_RECT_ANALYTIC = 1u, MTR_RECT_FLIP_NORMALS = 2u };     /* mtr_shape.is_rectangle */
typedef struct mtr_shape {          /* one scene shape = a contiguous triangle range */
    uint32_t first_tri, n_tris;
    uint32_t is_rectangle;          /* analytic `rectangle`: the shape is ONE primitive — mitsuba's Rectangle::ray_intersect
                                       (ray to object space, t = -o.z/d.z, |x|,|y| <= 1), one shading frame, sample_position by
                                       to_world.  Its two triangles (0,1,2) (0,2,3) of the corners (-1,-1) (1,-1) (1,1) (-1,1)
                                       only CARRY the material / emitter / index of the primitive (hits report the first one).
                                       Bit MTR_RECT_FLIP_NORMALS (ABI 9): the shape's `flip_normals` — geometric and shading
                                       normal are -normalize(du x dv) (and t = n x s follows); the local parameterisation,
                                       hence prim_uv and sample_position, is unchanged. */
    float    center[3], du[3], dv[3];   /* rectangle only: to_world*(0,0,0), half edges to_world*(1,0,0) - center, to_world*(0,1,0) - center */
    uint32_t has_to_world;          /* meshes: to_world below is the shape's object -> world transform (cube: of [-1,1]^3; obj / ply:
                                       of the file's coordinates).  Only an acceleration hint (oriented bounds); 0 = unknown */
    float    to_world[12];          /* row-major 3 x 4 affine */
} mtr_shape;

#define MTR_NLOS_NO_RELAY 0xffffffffu
typedef struct mtr_nlos_desc {
    float    sensor_origin[3];      /* nlos_capture_meter.sensor_origin (nloscapturemeter.py:103-105)     */
    uint32_t relay_shape;           /* index into shapes[]: the rectangle the nlos_capture_meter is attached to;
                                       MTR_NLOS_NO_RELAY: the sensor is the scene's `perspective` camera
                                       (mtr_scene_desc.camera), as in examples/transient-nlos/nlos-z-simple.xml      */
    float    laser_to_world[16];    /* projector world transform, row-major (after nlos.focus_emitter_*)   */
    float    laser_fov;             /* degrees, along x                                                    */
    float    laser_irradiance[3];   /* constant `irradiance` texture                                       */
    float    laser_scale;           /* `scale` (default 1)                                                 */
    uint32_t capture_type;          /* MTR_CAPTURE_*                                                       */
    uint32_t flags;                 /* MTR_NLOS_*                                                          */
    int32_t  filter_depth;          /* -1 = off (filter_bounces + 1 when that was given)                   */
    float    illumination_scan_fov; /* Exhaustive without FORCE_EQUAL_GRIDS: fov (degrees) of the laser grid scan
                                       (transientnlospath.py:346-376); the grid is film.laser_scan_width x _height */
    uint32_t n_shapes;
    const mtr_shape *shapes;        /* host; every triangle of the scene belongs to exactly one shape      */
    /* `is_confocal` capture meter (nloscapturemeter.py:111-119, :142): a 1 x 1 film whose every sensor ray goes to
     * nlos_capture_meter.laser_target (set by mitransient.nlos.focus_emitter_*) instead of the pixel-centre grid */
    uint32_t sensor_is_confocal;
    float    sensor_target[3];
} mtr_nlos_desc;

typedef struct mtr_scene_desc {
    uint32_t        n_tris;
    const float    *tri_verts;     /* host, n_tris*9 floats: p0 p1 p2, world space      */
    const uint32_t *tri_material;  /* host, n_tris: index into materials                */
    const int32_t  *tri_emitter;   /* host, n_tris: index into emitters or -1           */
    uint32_t        n_materials;
    const mtr_material *materials; /* host                                              */
    uint32_t        n_emitters;
    const mtr_emitter  *emitters;  /* host                                              */
    mtr_camera      camera;         /* with nlos != NULL: used only when nlos->relay_shape == MTR_NLOS_NO_RELAY */
    mtr_film_desc   film;
    const mtr_nlos_desc *nlos;      /* NULL: `perspective` sensor + `transient_path`; else the NLOS tier */
    /* Shape table (host; optional): shapes tile the triangle array in order.  Needed for analytic rectangles
     * (mtr_shape.is_rectangle); without it every triangle is a plain mesh triangle.  With nlos != NULL it must be the
     * same table as nlos->shapes. */
    uint32_t        n_shapes;
    const mtr_shape *shapes;
    /* Per-corner texture coordinates (host, n_tris*6 floats: uv0 uv1 uv2; optional).  They only fix the tangent of the
     * shading frame, as in mitsuba's Mesh::compute_surface_interaction: dp_du from the UV parameterisation when it is
     * not degenerate, coordinate_system(n) otherwise (and for every triangle when this is NULL); then
     * SurfaceInteraction::initialize_sh_frame: s = normalize(dp_du - n * dot(n, dp_du)), t = n x s. */
    const float    *tri_uv;
    /* Per-corner SHADING normals (host, n_tris*9 floats: n0 n1 n2, unit length, world space; optional; ABI 8).  A triangle
     * whose nine floats are all zero — and every triangle when this is NULL — is flat-shaded (face_normals = true, cube,
     * rectangle).  Otherwise, as in mitsuba's Mesh::compute_surface_interaction: sh_frame.n = normalize(b0 n0 + b1 n1 +
     * b2 n2), the tangent from dp_du by initialize_sh_frame; the geometric normal (si.n) keeps the ray offsets.  Meshes that are
     * SAMPLED (area emitters, NLOS hidden geometry) interpolate the same normals in Mesh::sample_position. */
    const float    *tri_normals;
    /* Bitmap textures referenced by mtr_material.albedo_texture (host; optional).  The lookup uses tri_uv exactly as given:
     * uv = fmadd(uv2, b2, fmadd(uv1, b1, uv0 b0)) (a rectangle: (prim_uv + 1) / 2); mitsuba's OBJ loader flips v
     * (flip_tex_coords), the caller passes the flipped coordinates. */
    uint32_t        n_textures;
    const mtr_texture *textures;
} mtr_scene_desc;

/* ---- integrator: `transient_path` properties (common.py:22-30) ---------- */
enum { MTR_FLAG_CAMERA_UNWARP = 1u,        /* common.py:25, transientpath.py:133-138 */
       MTR_FLAG_DISCARD_DIRECT_LIGHT = 2u, /* common.py:27, transientpath.py:173-176 */
       MTR_FLAG_FILM_ZERO = 4u,            /* caller guarantees that the film rows of the rendered pixels are
                                              all-zero on entry (first pass after TransientImageBlock.clear):
                                              the row flush may store instead of read-modify-write            */
       MTR_FLAG_KEEP_COUNTERS = 16u,       /* do not reset the context's device counters at the start of this call: a render
                                              issued as several asynchronous calls (row bands on alternating streams) sums
                                              its counters on the device; read them once with mtr_counters_read()        */
       MTR_FLAG_DETERMINISTIC = 32u,       /* fused kernel: accumulate the LDS rows (and the steady sums) in signed 2^-42 fixed
                                              point with 64-bit integer atomics instead of f32 atomics: sums no longer depend
                                              on the order in which lanes add, so two runs give the same bits (the wavefront
                                              organisation's scatter kernel always works this way).  Twice the LDS per row.  */
       MTR_FLAG_DEVELOPED_ROWS = 64u,      /* single-pass film lifecycle (ABI 9): `transient_hwt4` of mtr_render is the DEVELOPED tensor
                                              (H, W, T, 3) — for every pixel of [pixel_begin, pixel_end) the whole row is STORED,
                                              zeros included — so the call replaces mtr_film_clear + mtr_render +
                                              mtr_film_develop of the transient tensor (the weight channel of the reference's
                                              raw block stays 0: develop divides by 1, the developed values ARE the raw sums).
                                              The row holds the samples of THIS call only: use it when one call renders all
                                              samples of its pixels.  Honoured by the fused organisation with LDS rows
                                              (mtr_render_plan reports it); MTR_ERR_UNSUPPORTED otherwise.               */
       MTR_FLAG_PCG_INITSEQ_PLUS_LANE = 8u /* sampler seeding variant: PCG32 initseq = TEA.v1 + lane instead of TEA.v1.
                                              drjit's PCG32::seed(size, initstate, initseq) adds arange(size) to initseq;
                                              mitsuba's independent sampler passes size = 1 after the TEA scramble in the
                                              versions we know [upstream-unverified, SURVEY A.9].  Kept so that ONE real
                                              reference render decides the question without a code change
                                              (tests/test_reference_golden.py).                                */,
       MTR_FLAG_PCG_TEA64 = 128u           /* a third reading of the same call site (round 5; a new flag bit, no ABI change): m_rng.seed(1, sample_tea_64(seed, idx),
                                              sample_tea_64(idx, seed)) — 64-bit state and stream words, each the two TEA
                                              outputs glued together (v0 + (v1 << 32)) — instead of the two halves of one
                                              sample_tea_32(seed, idx) [upstream-unverified].  Wins over PLUS_LANE if both are set. */ };

/* which kernel organisation executes the path */
enum { MTR_MODE_AUTO = 0,
       MTR_MODE_FUSED = 1,      /* one persistent launch per tile; per-pixel LDS time histogram */
       MTR_MODE_WAVEFRONT = 2   /* per-bounce launches over SoA path queues in HBM + splat
                                   records + the stand-alone time-bin scatter-add kernel        */ };

typedef struct mtr_render_params {
    uint32_t spp_total;     /* samples per pixel of the WHOLE render: sample_scale = 1/spp_total
                               (common.py:173-175) and lane = pixel*spp_total + s             */
    uint32_t spp_begin;     /* this call renders samples s in [spp_begin, spp_end) ...         */
    uint32_t spp_end;
    uint32_t pixel_begin;   /* ... of crop-window pixels [pixel_begin, pixel_end) (row-major)  */
    uint32_t pixel_end;
    uint32_t seed;          /* mi.render(seed=) + sampler base seed (common.py:52)             */
    int32_t  max_depth;     /* -1 = unbounded                                                   */
    int32_t  rr_depth;
    uint32_t flags;         /* MTR_FLAG_*                                                       */
    uint32_t mode;          /* MTR_MODE_*                                                       */
    uint32_t spp_scale;     /* 0, or the sample count of the WHOLE multi-pass render (common.py:56-85: above 2^32 lanes
                               the reference renders passes of their own sampler each — lanes of a pass are indexed with
                               spp_total = the PASS's samples — while sample_scale stays 1/total_spp, :173-175)       */
    uint32_t reserve_cus;   /* (ABI 10) fused organisation: leave this many compute units WITHOUT a resident workgroup of the
                               persistent path kernel (grid = (CUs - reserve_cus) x workgroups per CU; at least one CU is
                               used).  The kernel otherwise owns every CU's LDS for the whole launch, so kernels of other
                               streams — RCCL's reduce-scatter of the previous row band — could only start between two
                               launches.  0 = use every CU (one GPU; the measured cost of 8 / 16 is in DESIGN.md section 7) */
    /* BAND COMPLETION WORDS (round 5; the struct's former reserved words: no ABI change).  n_bands > 0, fused organisation only
       (MTR_ERR_UNSUPPORTED otherwise: ask mtr_render_plan first): the pixel range is split into n_bands equal contiguous bands
       (the last takes the remainder) and, when every pixel of band b has been flushed to the film, the kernel stores band_epoch to
       band_done[b] with a system-scope release — rows and steady sums of the band are then visible to whatever waits for that
       word (hipStreamWaitValue32 on another stream: a multi-GPU caller starts band b's film reduction while the SAME launch still
       renders band b + 1, instead of issuing one launch per band).  The caller owns band_done (device memory, n_bands words) and
       picks an epoch the words do not hold yet.                                                                        */
    uint32_t n_bands;
    uint32_t band_epoch;
    uint64_t band_done;     /* device pointer (uint32_t *), or 0 */
} mtr_render_params;

/* in-kernel counters */
typedef struct mtr_counters {
    uint64_t paths;
    uint64_t rays_closest;
    uint64_t rays_shadow;
    uint64_t splats_issued;   /* in-range, non-zero time-bin contributions                      */
    uint64_t bounces;         /* loop iterations executed                                       */
    uint64_t splats_overflow; /* wavefront mode: records that took the global-atomic fallback   */
    uint64_t reserved[2];
} mtr_counters;

/* one time-resolved contribution, as consumed by the stand-alone scatter-add */
typedef struct mtr_splat_soa {
    const uint32_t *pixel;   /* device, n: y*W + x (film coordinates)                          */
    const float    *opl;     /* device, n: optical path length                                 */
    const float    *r, *g, *b; /* device, n: value already multiplied by sample_scale          */
    uint64_t        n;
    const uint32_t *laser;   /* device, n, or NULL (= 0): laser_x*laser_scan_height + laser_y of an
                                exhaustive_scan film (add_transient_data's laser_x / laser_y)    */
} mtr_splat_soa;

/* per-kernel timing of the last mtr_render (HIP events on the context stream) */
typedef struct mtr_kernel_times {
    float    total_ms;        /* whole mtr_render call on the stream                            */
    float    trace_ms;        /* fused: the path kernel | wavefront: sum of bounce kernels      */
    float    scatter_ms;      /* wavefront: sum of time-bin scatter-add launches (0 if fused)   */
    uint32_t trace_launches;
    uint32_t scatter_launches;
    float    wf_trace_ms;     /* wavefront (ABI 9): sum of the k_wf_trace launches alone (closest-hit and any-hit runs) */
    uint32_t wf_trace_kernel_launches; /* ... and their number                                   */
    float    wf_shade_ms;     /* wavefront (ABI 10): sum of the k_wf_shade launches (HBM-bound for scenes in HBM) */
} mtr_kernel_times;

typedef struct mtr_ctx   mtr_ctx;
typedef struct mtr_scene mtr_scene;

/* ------------------------------------------------------------------------ */
int  mtr_abi_version(void);

/* Context = one GPU + one stream.  device_ordinal >= 0 is required: this
 * library has no CPU path (MTR_ERR_NO_DEVICE when HIP sees no device). */
int  mtr_ctx_create(int device_ordinal, mtr_ctx **out);
void mtr_ctx_destroy(mtr_ctx *);
int  mtr_ctx_set_stream(mtr_ctx *, void *hip_stream);
const char *mtr_last_error(const mtr_ctx *);   /* never NULL; ctx may be NULL for creation errors */

/* Replaces mi.load_dict(scene_dict) for the supported subset (reference
 * call site: README.md:159, utils.py:78-220).  Copies the arrays, builds the
 * BVH2 on the host and uploads it. */
int  mtr_scene_create(mtr_ctx *, const mtr_scene_desc *, mtr_scene **out);
void mtr_scene_destroy(mtr_scene *);
/* transient_hdr_film traverse()/parameters_changed (transient_hdr_film.py:295-311):
 * change T / start / width between renders. */
int  mtr_scene_set_film(mtr_scene *, const mtr_film_desc *);
/* NLOS tier: (re)derive the laser / relay-wall / hidden-geometry tables and the scanned points
 * (TransientNLOSPath.prepare, transientnlospath.py:251-383) after mitransient.nlos.focus_emitter_*
 * (nlos.py:5-70) or an integrator property changed.  The shape table must match the scene's triangles. */
int  mtr_scene_set_nlos(mtr_scene *, const mtr_nlos_desc *);
/* BVH statistics for tests: nodes, max depth, leaf count. */
int  mtr_scene_bvh_info(const mtr_scene *, uint32_t *n_nodes, uint32_t *max_depth, uint32_t *n_leaves);
/* Which specialised kernels the scene's tables select (for tests and tools; no counterpart in the reference, whose tracing
 * JIT specialises on the scene implicitly): MTR_TRAIT_* bits. */
#define MTR_TRAIT_DIFFUSE          1u   /* every material plain one-sided diffuse */
#define MTR_TRAIT_ONE_RECT_EMITTER 2u   /* exactly one emitter, an analytic rectangle */
#define MTR_TRAIT_LEAF_PAIR        4u   /* no leaf of the LDS-staged tree beyond one triangle pair */
#define MTR_TRAIT_FLAT_TOP         8u   /* top level = rectangles, triangle leaves and box nodes: the fused kernel does not walk a tree */
#define MTR_TRAIT_FLAT_LEAVES     16u   /* ... and there are triangle leaves among them */
#define MTR_TRAIT_NO_LOBES        32u   /* extended shading (interpolated normals, bitmaps) without any microfacet lobe / plastic / thin dielectric */
#define MTR_TRAIT_GREY            64u   /* every colour (materials, emitters, the NLOS laser) has three equal channels, no bitmaps: r == g == b in every contribution */
int  mtr_scene_traits(const mtr_scene *, uint32_t *traits);

/* TransientImageBlock.clear (transient_image_block.py:56-70): zero the
 * (H,W,T,4) f32 accumulator and the (H,W,4) steady accumulator. */
int  mtr_film_clear(mtr_ctx *, const mtr_film_desc *, float *transient_hwt4 /*device, may be NULL*/,
                    float *steady_hw4 /*device, may be NULL*/);

/* TransientADIntegrator.render pass (common.py:157-210) + TransientPath.sample
 * (transientpath.py:88-326) + add_transient_data/put_/accum
 * (transient_hdr_film.py:250-276, transient_image_block.py:103-151):
 * ADDS the contributions of the requested lanes into
 *   transient_hwt4 : device f32 (H, W, T, 4)  channels R,G,B,W (W stays 0)
 *   steady_hw4     : device f32 (H, W, 4)     sum of L over samples, and the sample count in .w
 * counters/times may be NULL.  Asynchronous on the context stream unless
 * counters or times are requested (then it synchronises the stream). */
int  mtr_render(mtr_scene *, const mtr_render_params *,
                float *transient_hwt4, float *steady_hw4,
                mtr_counters *counters_out /*host*/, mtr_kernel_times *times_out /*host*/);

/* Which kernel organisation mtr_render would run for these parameters on this scene (MTR_MODE_AUTO resolved exactly as
 * mtr_render resolves it; MTR_MODE_FUSED or MTR_MODE_WAVEFRONT in *mode_out).  A caller that overlaps several mtr_render
 * calls of ONE scene on different streams needs this: only the fused organisation keeps its per-launch state apart
 * (rotating work tickets); the wavefront organisation has one workspace per scene and must stay on one stream. */
int  mtr_render_plan(mtr_scene *, const mtr_render_params *, uint32_t *mode_out,
                     uint32_t *developed_rows_ok /* may be NULL: 1 when MTR_FLAG_DEVELOPED_ROWS would be honoured */);

/* Zero the context's device counters on the context stream (then issue every mtr_render of the render with
 * MTR_FLAG_KEEP_COUNTERS and read the sums once with mtr_counters_read). */
int  mtr_counters_reset(mtr_ctx *);

/* The context's device counters as they stand (summed over every mtr_render since the last one without
 * MTR_FLAG_KEEP_COUNTERS).  The caller has synchronised the streams those renders ran on. */
int  mtr_counters_read(mtr_ctx *, mtr_counters *out /*host*/);

/* TransientHDRFilm.develop / develop_transient_ (transient_hdr_film.py:210-248)
 * and steady.develop (common.py:206,212):  raw (H,W,T,4) -> (H,W,T,3) with the
 * weight division (w==0 -> 1), steady (H,W,4) -> (H,W,3) = sum / count. */
int  mtr_film_develop(mtr_ctx *, const mtr_film_desc *,
                      const float *transient_hwt4, float *transient_hwt3 /*device, may be NULL*/,
                      const float *steady_hw4, float *steady_hw3 /*device, may be NULL*/);

/* The time-bin scatter-add alone: add_transient_data + put_ + accum
 * (transient_hdr_film.py:263-276, transient_image_block.py:131-149) over n
 * splats.  variant 0 = global f32 atomics (the contract form: one atomic per channel wherever the contribution lands).
 * variant 1 = LDS-privatised rows, one workgroup per pixel: input whose `pixel` is non-decreasing (the order of the
 * reference's own lanes, pixel * spp + s) is consumed as it is; any other order is first partitioned by pixel on the
 * device (ABI 10; two scatter passes over 16-byte records — needs 32 bytes of device workspace per contribution for the
 * duration of the call and synchronises the stream once; films whose pixel index and row position do not fit one 32-bit key
 * fall back to variant 0 on the device).  OR MTR_SPLAT_FILM_ZERO into `variant` when the film is all-zero on entry: the row
 * flush then stores whole rows instead of read-modify-write. */
#define MTR_SPLAT_FILM_ZERO 0x100
int  mtr_splat_add(mtr_ctx *, const mtr_splat_soa *, const mtr_film_desc *, int variant,
                   float *transient_hwt4 /*device*/, float *elapsed_ms /*host, may be NULL*/);

/* Release the device workspaces the context keeps between calls (ABI 10: the partition workspace of mtr_splat_add on
 * unsorted input — 32 bytes per contribution; a workspace above 256 MiB is released by the call itself, a smaller one
 * stays with the context).  Synchronises the context's stream. */
int  mtr_ctx_trim(mtr_ctx *);

/* Debug/test aid: per-lane splat log of one render (records of 8 x u32:
 * lane, depth|kind<<16, pixel, bin, r,g,b bits, opl bits), capacity in records.
 * Pass NULL to disable. The count is written to *n_records_device (u64). */
int  mtr_debug_set_splat_log(mtr_scene *, uint32_t *log_device, uint64_t capacity, uint64_t *n_records_device);

#ifdef __cplusplus
}
#endif
#endif /* MITRANSIENT_AMD_H */
