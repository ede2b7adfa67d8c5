// mtr_core.h — per-path arithmetic of the transient path tracer (PRODUCT code).
//
// Everything a single path does between two queue/LDS accesses lives here as
// force-inlined functions: PCG32/TEA sampler, perspective ray generation, BVH2
// traversal over node packets, Moller-Trumbore, shading frame, BSDFs, rectangle
// area-emitter sampling, MIS, Russian roulette and the OPL -> time-bin mapping.
// The kernels in mtr_kernels.hip only add the CDNA4 machinery around it (LDS
// staging, wave64 compaction, queues, histograms).
//
// Reference map (mitransient/integrators/transientpath.py):
//   path_begin()  :118-138 + ADIntegrator.sample_rays (common.py:159)
//   path_bounce() :140-319, one loop iteration
//   film_bin()    films/transient_hdr_film.py:263-265 + render/transient_image_block.py:131-146
//
// Numerics contract: IEEE f32, compiled with -ffp-contract=off, fmaf only
// where written, correctly rounded 1/x and sqrtf, polynomial sin/cos.  The file is
// plain C++ so tests can also compile it for the host (tests/host_harness.cpp).
#pragma once
#ifndef MTR_WIDE
#define MTR_WIDE 8          // children per wide node (even, <= 16)
#endif

#include <stdint.h>
#include <math.h>
#include "../../include/mitransient_amd.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MTR_HD __host__ __device__ __forceinline__
#else
#define MTR_HD inline __attribute__((always_inline))
#endif

namespace mtr {

struct f3 { float x, y, z; };

MTR_HD f3 mk(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
MTR_HD f3 operator-(f3 a, f3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
MTR_HD f3 operator-(f3 a) { return mk(-a.x, -a.y, -a.z); }
MTR_HD f3 operator*(f3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
// vector / scalar is ONE correctly rounded reciprocal and three multiplies (numerics contract)
MTR_HD f3 operator/(f3 a, float s) { float r = 1.0f / s; return mk(a.x * r, a.y * r, a.z * r); }
MTR_HD float dot(f3 a, f3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
MTR_HD f3 cross(f3 a, f3 b)
{
    return mk(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}
MTR_HD f3 normalize(f3 a) { return a / sqrtf(dot(a, a)); }
MTR_HD f3 fma3(f3 a, float s, f3 b) { return mk(fmaf(a.x, s, b.x), fmaf(a.y, s, b.y), fmaf(a.z, s, b.z)); }
MTR_HD float max3(float a, float b, float c) { return fmaxf(a, fmaxf(b, c)); }
MTR_HD bool sign_neg(float s) { return s < 0.0f || (s == 0.0f && signbit(s)); }

constexpr float kPi = 3.14159265358979323846f;
constexpr float kInvPi = 0.31830988618379067154f;
constexpr float kRayEps = 1500.0f * 5.9604644775390625e-8f;
constexpr float kShadowEps = kRayEps * 10.0f;
constexpr float kInf = __builtin_huge_valf();
constexpr float kEdgeEps = 1.9073486328125e-06f;     // 2^-19: how far a barycentric coordinate may undershoot a triangle edge (trav_leaf_test)

// ---------------------------------------------------------------- sampler
struct Rng { uint64_t state, inc; };

MTR_HD uint32_t rng_u32(Rng &r)
{
    uint64_t old = r.state;
    r.state = old * 0x5851f42d4c957f2dULL + r.inc;
    uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u);
    uint32_t rot = (uint32_t)(old >> 59u);
    return (xs >> rot) | (xs << ((0u - rot) & 31u));
}
MTR_HD float rng_f32(Rng &r)
{
    uint32_t u = (rng_u32(r) >> 9) | 0x3f800000u;
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f - 1.0f;
}
// TEA-scrambled (seed, lane) -> PCG32 stream; `inc` is recomputable from (seed, lane)
MTR_HD void tea4(uint32_t &v0, uint32_t &v1)
{
    uint32_t sum = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        sum += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + sum) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + sum) ^ ((v0 >> 5) + 0x7e95761eu);
    }
}
// The (initstate, initseq) pair mitsuba's independent sampler hands to PCG32::seed for (seed, lane) — three readings of one
// upstream call site, selected by render flags (include/mitransient_amd.h) until a real reference render decides:
//   default                         sample_tea_32(seed, lane) -> (v0, v1)
//   MTR_FLAG_PCG_INITSEQ_PLUS_LANE  ... with lane added to the stream word (drjit's PCG32::seed adds arange(size))
//   MTR_FLAG_PCG_TEA64              (sample_tea_64(seed, lane), sample_tea_64(lane, seed)), sample_tea_64 = v0 + (v1 << 32)
MTR_HD void rng_seed_words(uint32_t seed, uint32_t lane, uint32_t flags, uint64_t &initstate, uint64_t &initseq)
{
    uint32_t v0 = seed, v1 = lane;
    tea4(v0, v1);
    if (flags & MTR_FLAG_PCG_TEA64) {
        uint32_t w0 = lane, w1 = seed;
        tea4(w0, w1);
        initstate = (uint64_t)v0 + ((uint64_t)v1 << 32); initseq = (uint64_t)w0 + ((uint64_t)w1 << 32);
    } else {
        initstate = (uint64_t)v0;
        initseq = (uint64_t)v1 + ((flags & MTR_FLAG_PCG_INITSEQ_PLUS_LANE) ? (uint64_t)lane : 0ull);
    }
}
MTR_HD uint64_t rng_inc_of(uint32_t seed, uint32_t lane, uint32_t flags = 0u)
{
    uint64_t st, sq;
    rng_seed_words(seed, lane, flags, st, sq);
    return (sq << 1u) | 1u;                                  // == rng_seed(...).inc
}
MTR_HD Rng rng_seed(uint32_t seed, uint32_t lane, uint32_t flags = 0u)
{
    uint64_t st, sq;
    rng_seed_words(seed, lane, flags, st, sq);
    Rng r;
    r.inc = (sq << 1u) | 1u;
    r.state = 0u;
    rng_u32(r);
    r.state += st;
    rng_u32(r);
    return r;
}

// ---------------------------------------------------------------- warps
MTR_HD void sincos_quarter(float x, float &s, float &c)   // |x| <= pi/4
{
    float z = x * x;
    float ps = fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
    s = fmaf(x * z, ps, x);
    float pc = fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
    c = fmaf(z * z, pc, fmaf(-0.5f, z, 1.0f));
}
// One frequency of a phasor-film contribution (phasor_image_block.py:49-56): phase = fmod(-2 pi f opl, 2 pi) with
// fmod(x, y) = x - y * floor(x / y), all f32; cos / sin by reduction to a multiple of pi/2 (two-constant Cody-Waite)
// and the quarter-range polynomials above (numerics contract; Dr.Jit's own sincos is not reproducible bit for bit).
MTR_HD void phasor_term(float freq, float opl, float &c, float &s)
{
    const float x = (-6.283185307179586f * freq) * opl, y = 6.283185307179586f;
    const float phase = x - y * floorf(x / y);
    const float k = floorf(fmaf(phase, 0.6366197723675814f, 0.5f));          // nearest multiple of pi/2
    float r = fmaf(-k, 1.5707962512969971f, phase);                             // pi/2 = hi + lo
    r = fmaf(-k, 7.549789415861596e-08f, r);
    float sq, cq;
    sincos_quarter(r, sq, cq);
    const uint32_t q = (uint32_t)(int32_t)k & 3u;
    s = (q == 0u) ? sq : (q == 1u) ? cq : (q == 2u) ? -sq : -cq;
    c = (q == 0u) ? cq : (q == 1u) ? -sq : (q == 2u) ? -cq : sq;
}

// [mitsuba3: warp::square_to_uniform_disk_concentric]
MTR_HD void concentric_disk(float u1, float u2, float &px, float &py)
{
    float x = fmaf(2.0f, u1, -1.0f), y = fmaf(2.0f, u2, -1.0f);
    bool swap = fabsf(x) < fabsf(y);
    float r = swap ? y : x, rp = swap ? x : y;
    float phi = (0.25f * kPi) * rp / r;
    if (x == 0.0f && y == 0.0f) phi = 0.0f;
    float s, c;
    sincos_quarter(phi, s, c);
    float cs = swap ? s : c, sn = swap ? c : s;
    px = r * cs; py = r * sn;
}
MTR_HD f3 cosine_hemisphere(float u1, float u2)
{
    float px, py;
    concentric_disk(u1, u2, px, py);
    float zz = 1.0f - fmaf(px, px, py * py);
    return mk(px, py, sqrtf(zz > 0.0f ? zz : 0.0f));
}

// ---------------------------------------------------------------- scene (device layout)
// 16-byte quads: every record below is read as whole quads (ds_read_b128 from LDS,
// global_load_dwordx4 from HBM/L2).
struct alignas(16) q4 { float x, y, z, w; };
// two independent f32 lanes of one register pair: v_pk_fma_f32 on the device (same roundings as two fmaf)
struct f2 { float x, y; };
#if defined(__HIP_DEVICE_COMPILE__)
typedef float v2f_ __attribute__((ext_vector_type(2)));
MTR_HD f2 mul2(f2 a, f2 b) { const v2f_ r = v2f_{ a.x, a.y } * v2f_{ b.x, b.y }; return f2{ r.x, r.y }; }
MTR_HD f2 mul2(f2 a, float b) { const v2f_ r = v2f_{ a.x, a.y } * v2f_{ b, b }; return f2{ r.x, r.y }; }
MTR_HD f2 rsub2(float a, f2 b) { const v2f_ r = v2f_{ a, a } - v2f_{ b.x, b.y }; return f2{ r.x, r.y }; }          // a - b
MTR_HD f2 add2(f2 a, f2 b) { const v2f_ r = v2f_{ a.x, a.y } + v2f_{ b.x, b.y }; return f2{ r.x, r.y }; }
MTR_HD f2 fma2(f2 a, f2 b, f2 c) { const v2f_ r = __builtin_elementwise_fma(v2f_{ a.x, a.y }, v2f_{ b.x, b.y }, v2f_{ c.x, c.y }); return f2{ r.x, r.y }; }
MTR_HD f2 fma2(f2 a, float b, f2 c) { const v2f_ r = __builtin_elementwise_fma(v2f_{ a.x, a.y }, v2f_{ b, b }, v2f_{ c.x, c.y }); return f2{ r.x, r.y }; }
#else
MTR_HD f2 mul2(f2 a, f2 b) { return f2{ a.x * b.x, a.y * b.y }; }
MTR_HD f2 mul2(f2 a, float b) { return f2{ a.x * b, a.y * b }; }
MTR_HD f2 rsub2(float a, f2 b) { return f2{ a - b.x, a - b.y }; }
MTR_HD f2 add2(f2 a, f2 b) { return f2{ a.x + b.x, a.y + b.y }; }
MTR_HD f2 fma2(f2 a, f2 b, f2 c) { return f2{ fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y) }; }
MTR_HD f2 fma2(f2 a, float b, f2 c) { return f2{ fmaf(a.x, b, c.x), fmaf(a.y, b, c.y) }; }
#endif
MTR_HD f2 neg2(f2 a) { return f2{ -a.x, -a.y }; }
MTR_HD f2 fma2(f2 a, float b, float c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef float v2f __attribute__((ext_vector_type(2)));
    const v2f r = __builtin_elementwise_fma(v2f{ a.x, a.y }, v2f{ b, b }, v2f{ c, c });
    return f2{ r.x, r.y };
#else
    return f2{ fmaf(a.x, b, c), fmaf(a.y, b, c) };
#endif
}
MTR_HD uint32_t fbits(float f) { uint32_t u; __builtin_memcpy(&u, &f, 4); return u; }
MTR_HD float bitsf(uint32_t u) { float f; __builtin_memcpy(&f, &u, 4); return f; }

// BVH2 "node packet": both children's boxes + refs in one 64-byte record, child-interleaved so
// that the two slab tests run as packed pairs:
//   q[0] = (lo0.x, lo1.x, hi0.x, hi1.x)   q[1] = (.y ...)   q[2] = (.z ...)   q[3] = (c0, c1, -, -)
// ref >= 0: inner node; ref < 0: leaf, ~ref = (first_tri << 2) | (count - 1), count in 1..4.
// An absent child has an inverted box (lo = +inf, hi = -inf).
struct alignas(16) Node {
    q4 q[4];
};
MTR_HD void node_set_child(Node &n, int c, const float *lo, const float *hi, int32_t ref)
{
    float *f = &n.q[0].x;
    for (int k = 0; k < 3; ++k) { f[4 * k + c] = lo[k]; f[4 * k + 2 + c] = hi[k]; }
    f[12 + c] = bitsf((uint32_t)ref);
}
// Wide node (scenes staged in LDS): up to kWide children, stored as child PAIRS in the layout of a BVH2 packet so that
// the slab tests of a pair run as packed math: box[3j + k] = (lo_a, lo_b, hi_a, hi_b) of axis k for children a = 2j, b = 2j + 1.
// Children are sorted by centroid along `axis`; a ray walks them in index order, or in reverse when its direction is
// negative on that axis.  Refs as in Node (>= 0 wide node, < 0 leaf); an absent child has an inverted box.
// A BVH2 over a few large quads (the walls of a room) cannot separate them — every ray visits all their ancestors;
// one 8-wide step replaces up to seven of those dependent 2-wide steps and, more important on a 64-lane wave, brings the
// per-ray step count of different lanes close together (Cornell box: wave-steps at 19 % lane utilisation with BVH2).
// Rectangle children come first (indices 0 .. n_quads-1) and are always visited first: their test is cheap and usually
// yields the final hit of a ray inside a room, and every lane of a wave then runs its rectangle tests at the same time.
// OBJECT nodes (flags & 1): the children's boxes are expressed in the object space of one small mesh shape (`xf` = rows of
// the world -> object affine map), where e.g. the faces of a rotated `cube` are flat and axis-aligned, so the slab tests
// cull exactly instead of through fat world-space boxes.  The ray parameter t is invariant under affine maps: entry / exit
// distances computed in object space compare directly with tmax and the closest hit.  All children of an object node are
// leaves (their triangles are tested in WORLD space as everywhere else: the hit arithmetic never changes).
template <uint32_t W>
struct alignas(16) WNodeT {
    q4 box[3 * W / 2];
    int32_t ref[W];
    uint32_t axis, count, n_quads, flags;     // n_quads: children 0 .. n_quads-1 are rectangle leaves; flags bit 0: object node,
                                              // bit 1: BOX node — the object is the cube [-1,1]^3 and child f = 2 * axis +
                                              // (coordinate = +1) is the leaf with the two triangles of that face (box_select)
    float xf[12];                             // object nodes: (R | T) rows, local = R * p + T
    static constexpr uint32_t kBytes = 28u * W + 64u, kRefOff = 24u * W, kHdrOff = 28u * W, kXfOff = 28u * W + 16u;
    static constexpr uint32_t kMask = (1u << W) - 1u, kRevBit = 1u << W, kQuadShift = W + 1u, kNodeShift = W + 5u;
};
constexpr uint32_t kWide = MTR_WIDE;
typedef WNodeT<kWide> WNode;
// FLAT TOP LEVEL (round 6; scene trait kTrFlatTop): the root's children are primitives — analytic rectangles, triangle leaves — and
// BOX nodes only: a room with a few cubes in it (the Cornell box of configs 1-3), a relay wall and a handful of triangles (config 4).  Such a scene is not walked at all (flat_walk_device): no stack, no
// group words, no votes between node and primitive steps.  The boxes' world -> object rows are handed to the kernels in
// their ARGUMENT, so that the object-space transform and the face selection of a box run on scalar operands (s_load from the
// kernarg segment) for every lane at once.  The box nodes are wnodes[node0 .. node0 + n_boxes).
constexpr uint32_t kFlatMaxBoxes = 4;
struct FlatTop {
    uint32_t n_boxes, node0, n_quads, prim_mask;      // prim_mask: the root's children that are PRIMITIVES (rectangles: bits 0 .. n_quads-1; triangle leaves)
    float xf[kFlatMaxBoxes][16];          // copies of WNode::xf (rows A, B, C), then S = (|A.x|+|A.y|+|A.z|, ... of B, ... of C, 0)
};
struct FlatHdr { uint32_t n_boxes, node0, n_quads, prim_mask; };
struct FlatXf { q4 A, B, C, S; };
// Scenes walked in HBM: 4-wide nodes with the children's boxes quantised to 8 bits per plane on the node's own grid
// (origin = the node's lower corner, one power-of-two step per axis) — 64 bytes, the size of a BVH2 packet, for twice the
// fan-out.  The trace kernel is bound by the number of 16-byte-per-lane loads it issues (divergent addresses: 0.6 - 0.8
// per clock per CU, profiles/r01_divergent_load_microbench.txt), not by the bytes behind them, so this halves its load
// count per ray; an uncompressed 4-wide node (128 B) halved the steps and doubled the loads per step — no gain (measured).
// Quantisation rounds lo down and hi up, so culling stays conservative; hits are decided by the triangle test alone.
//   q[0] = (org.x, org.y, org.z, meta)   meta = ex | ey << 8 | ez << 16 | axis << 24 | count << 26  (e*: biased exponents)
//   q[1] = refs   q[2] = (lo.x, lo.y, lo.z, hi.x)   q[3] = (hi.y, hi.z, -, -)     byte c of a plane word = child c
struct alignas(16) QNode4 { q4 q[4]; };
static_assert(sizeof(QNode4) == 64, "quantised 4-wide node");
// The same, 8 wide (96 bytes): the planes of the eight children, still 8 bits on the node's own grid, take 48 bytes, the
// references 32 (read one at a time, when a child is visited).  Fewer, fatter steps: a walk visits ~40 % fewer nodes.
//   q[0] = (org.x, org.y, org.z, meta)   meta = ex | ey << 8 | ez << 16 | axis << 24 | count << 26   (count 1 .. 8)
//   q[1] = (lo.x c0-3, lo.x c4-7, lo.y c0-3, lo.y c4-7)   q[2] = (lo.z c0-3, lo.z c4-7, hi.x c0-3, hi.x c4-7)
//   q[3] = (hi.y c0-3, hi.y c4-7, hi.z c0-3, hi.z c4-7)   q[4], q[5] = refs of children 0-3, 4-7
struct alignas(16) QNode8 { q4 q[6]; };
static_assert(sizeof(QNode8) == 96, "quantised 8-wide node");
// (padded to one 128-byte line per node — a 96-byte stride puts the 64 bytes a step reads across two lines for one node in
// four — config 5 is 3 % SLOWER, k_wf_trace 975 -> 1005 ms: the tree's footprint in L2 matters more than the line crossings)
// triangles, split by use and stored by SLOT: every leaf starts on an even slot and owns ceil(count / 2) pairs of slots
// (an odd leaf repeats its last triangle in the pad slot; the pad is never reported as a hit).
// Intersection record = one PAIR of slots with the two triangles interleaved, so that one 16-byte read delivers two
// (A, B) register pairs for the packed Moller-Trumbore of trav_leaf_step; edges precomputed (e = p - p0, the same f32
// subtraction Moller-Trumbore starts with):
//   g[0] = (p0.x A,B  p0.y A,B)  g[1] = (p0.z A,B  e1.x A,B)  g[2] = (e1.y A,B  e1.z A,B)  g[3] = (e2.x A,B  e2.y A,B)
//   g[4] = (e2.z A,B  orig A, orig B)     orig: index of the triangle in the caller's array (ties on t go to the lower one)
// A `rectangle` (analytic primitive) owns one pair of slots; its record holds the rows of to_object instead:
//   g[0] = (rz.x, rz.y, rz.z, tz)  g[1] = (rx.x, rx.y, rx.z, tx)  g[2] = (ry.x, ry.y, ry.z, ty)  g[4] = (-, -, orig, kQuadMark)
// and it is alone in its leaf, whose reference carries kLeafQuadBit.
// (round 5) `orig` is the TIE-BREAK WORD (original index << 3) | list key: ordered like the original index, and its low three bits
// are the material-type list the wavefront organisation files a hit on this triangle under (hit_list_key) — k_wf_trace reads the
// key off the hit it already holds instead of gathering shading record and material (two dependent loads per ray, -3.5 % on the kernel)
struct alignas(16) TriPair { q4 g[5]; };
// material type -> hit list of the wavefront organisation (kWfKeys: diffuse-like smooth lobes 0, conductor 1, dielectric 2, none 3; 4 = miss)
MTR_HD uint32_t hit_list_key(uint32_t type)
{
    if (type == MTR_BSDF_ROUGHCONDUCTOR || type == MTR_BSDF_ROUGHPLASTIC || type == MTR_BSDF_ROUGHDIELECTRIC || type == MTR_BSDF_PLASTIC) return 0u;   // the extended smooth lobes share the list of the smooth BSDFs (emitter sampling)
    if (type == MTR_BSDF_THINDIELECTRIC) return MTR_BSDF_DIELECTRIC;                  // two delta lobes: the dielectrics' list
    return type;
}
constexpr uint32_t kQuadMark = 0xffffffffu;
constexpr uint32_t kLeafQuadBit = 0x40000000u;      // in the leaf code ~ref = (first_slot << 2) | (count - 1)
// shading record per slot (5 quads): flat frame, the three vertices (hit point = barycentric blend), material | emitter
//   h[0] = (n.x, n.y, n.z, s.x)  h[1] = (s.y, s.z, t.x, t.y)  h[2] = (t.z, p1.x, p1.y, p1.z)  h[3] = (p2.x, p2.y, p2.z, p0.x)
//   h[4] = (p0.y, p0.z, mat_em, orig)      mat_em: material index | (emitter index + 1) << 16
// rectangle slots: p0 := to_world * (0,0,0), p1 := du, p2 := dv (hit point = c + du * u + dv * v), orig |= kShadeQuadBit
// smooth-shaded triangles (interpolated vertex normals, mtr_scene_desc.tri_normals): h[0] = (ng.x, ng.y, ng.z, dp_du.x),
// h[1] = (dp_du.y, dp_du.z, -, -) — the geometric normal and the tangent direction from which the frame is built at the hit
// (hit_ctx) — orig |= kShadeSmoothBit; the three vertex normals live in SceneView::vnormals[3 * slot ..]
#ifndef MTR_TSHADE_QUADS
#define MTR_TSHADE_QUADS 5      // (experiments: 8 = one record per 128-byte line; profiles/r05_trace_experiments.txt)
#endif
struct alignas(16) TriShade { q4 h[MTR_TSHADE_QUADS]; };
constexpr uint32_t kShadeQuadBit = 0x80000000u;
constexpr uint32_t kShadeSmoothBit = 0x40000000u;
struct alignas(16) Emitter {                       // 80 B
    float center[3], du[3], dv[3], n[3], radiance[3], inv_area;       // rectangle: analytic sampling
    uint32_t is_mesh, first_tri, n_tris, pad;                          // mesh: triangle range (ORIGINAL indices)
};

struct Camera {
    float s2c[16];
    float tw[16];
    float near_clip, far_clip;
};

struct Film {
    uint32_t width, height, crop_w, crop_h, crop_x, crop_y;
    uint32_t bins;       // length of a pixel's row: T, or Lw*Lh*T for an exhaustive_scan film (row = [laser][t])
    float start_opl, bin_width;
    uint32_t tbins;      // T: the time bins of one (pixel, laser) histogram
    uint32_t lasers;     // Lw*Lh (1 without exhaustive_scan)
    uint32_t n_freq;     // phasor_hdr_film: F > 0, the tensor is H x W x (2F + 1) and `bins` / `tbins` are 1
    const float *freq;   // [F]
};

struct SceneView {
    bool node_pairs;          // scene staged in LDS: fetch the entry / exit planes of a node by sign-dependent OFFSETS
    const Node *nodes;
    const WNode *wnodes;      // non-null: traverse the wide tree instead of `nodes` (scene staged in LDS)
    const QNode4 *wnodes4;    // non-null (and wnodes null): traverse the quantised 4-wide tree (scene in HBM)
    const QNode8 *wnodes8q;   // non-null (and wnodes null): traverse the quantised 8-wide tree instead (scene in HBM)
    const TriPair *tpairs;    // [n_slots / 2]
    const TriShade *tshade;
    const mtr_material *mats;
    const Emitter *ems;
    uint32_t n_emitters;
    uint32_t n_slots;         // triangle slots (even; >= the triangle count)
    // area sampling of triangle meshes (mesh emitters, NLOS hidden geometry), by ORIGINAL triangle index:
    // 3 quads (p0, e1.x) (e1.yz, e2.xy) (e2.z, n) and the face distribution normalised within the mesh
    const q4 *samp_tris;
    const q4 *samp_vn;        // vertex normals of the sampled meshes (null unless one has them): mesh_sample_position
    const float *face_pmf, *face_cdf;
    const q4 *vnormals;       // [3 * n_slots] vertex normals of smooth-shaded slots (HBM; null when every triangle is flat)
    // bitmap textures (HBM; null without textures): all texels as RGBA f32, per texture (first texel, width, height, -),
    // and the corner texture coordinates by slot: (u0, v0, u1, v1) (u2, v2, -, -)
    const q4 *texels; const q4 *tex_info; const q4 *uvs;
    uint32_t flat_off;        // kTrFlatTop kernels: byte offset of the scene's FlatTop inside the kernel's argument (kernarg_copy)
};

// [mitsuba3: DiscreteDistribution::sample_reuse_pmf] on a normalised f32 table
MTR_HD uint32_t distr_sample_reuse(const float *cdf, const float *pmf, uint32_t n, float value, float &reused, float &pmf_out)
{
    uint32_t i = 0;
    while (i + 1 < n && !(value < cdf[i])) ++i;
    while (i + 1 < n && pmf[i] == 0.0f) ++i;
    const float prev = i ? cdf[i - 1] : 0.0f;
    reused = (value - prev) / pmf[i];
    pmf_out = pmf[i];
    return i;
}

// [mitsuba3: Mesh::sample_position] face by area (reusing sample.y), then warp::square_to_uniform_triangle
// samp_vn (or null): three vertex normals per triangle, .w of the first = 1 on a mesh with vertex normals — then
// ps.n = normalize(fmadd(n0, 1 - b.x - b.y, fmadd(n1, b.x, n2 * b.y))) instead of the face normal
MTR_HD void mesh_sample_position(const q4 *samp_tris, const float *face_cdf, const float *face_pmf, uint32_t first_tri,
                                 uint32_t n_tris, float u1, float u2, f3 &p, f3 &n, const q4 *samp_vn = nullptr)
{
    float sy = u2;
    uint32_t fi = 0;
    if (n_tris > 1) { float r2, fp; fi = distr_sample_reuse(face_cdf + first_tri, face_pmf + first_tri, n_tris, u2, r2, fp); sy = r2; }
    const q4 *t = samp_tris + 3 * (size_t)(first_tri + fi);
    const q4 a = t[0], b = t[1], cc = t[2];
    const float tt = sqrtf(fmaxf(1.0f - u1, 0.0f));
    const float b0 = 1.0f - tt, b1 = tt * sy;
    p = mk(fmaf(a.w, b0, fmaf(b.z, b1, a.x)), fmaf(b.x, b0, fmaf(b.w, b1, a.y)), fmaf(b.y, b0, fmaf(cc.x, b1, a.z)));
    n = mk(cc.y, cc.z, cc.w);
    if (samp_vn) {
        const q4 *v = samp_vn + 3 * (size_t)(first_tri + fi);
        const q4 n0 = v[0];
        if (n0.w != 0.0f) {
            const q4 n1 = v[1], n2 = v[2];
            const float w0 = 1.0f - b0 - b1;
            n = normalize(mk(fmaf(n0.x, w0, fmaf(n1.x, b0, n2.x * b1)), fmaf(n0.y, w0, fmaf(n1.y, b0, n2.y * b1)),
                             fmaf(n0.z, w0, fmaf(n1.z, b0, n2.z * b1))));
        }
    }
}

// exact u32 division by an invariant divisor in 5 instructions (Granlund-Montgomery, the branch-free form):
//   q = (t + ((n - t) >> s1)) >> s2,  t = mulhi(m, n)        — the hardware has no integer divide (~35 instructions)
struct FastDiv { uint32_t m, s1, s2, d; };
inline FastDiv fastdiv_make(uint32_t d)
{
    FastDiv f; f.d = d ? d : 1u;
    if (f.d == 1u) { f.m = 0u; f.s1 = 0u; f.s2 = 0u; return f; }
    uint32_t l = 0; while ((1ull << l) < f.d) ++l;                       // ceil(log2 d)
    f.m = (uint32_t)((((1ull << l) - f.d) << 32) / f.d + 1ull);
    f.s1 = 1u; f.s2 = l - 1u;
    return f;
}
MTR_HD uint32_t fastdiv(uint32_t n, const FastDiv &f)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t t = __umulhi(f.m, n);
#else
    const uint32_t t = (uint32_t)(((uint64_t)f.m * n) >> 32);
#endif
    return (t + ((n - t) >> f.s1)) >> f.s2;
}

struct RenderConst {
    FastDiv div_crop_w;
    uint32_t spp_total;
    uint32_t seed;
    uint32_t max_depth;      // 0xffffffff = unbounded
    uint32_t rr_depth;
    uint32_t flags;
    float sample_scale;      // (float)(1.0 / spp_total), computed on the host
    float inv_crop_w, inv_crop_h, off_x, off_y;   // sample_rays: scale / offset of the film position
    float n_emitters_f, inv_n_emitters;
};

MTR_HD f3 ld3(const float *p) { return mk(p[0], p[1], p[2]); }

// ---------------------------------------------------------------- intersection
struct Hit { float t, u, v; int32_t prim; };
struct Ray { f3 o, d; float tmax; };

// reciprocal direction of the slab tests.  It only feeds CULLING (conservative: padded boxes, hits are decided by the
// primitive tests alone), so the device takes the hardware reciprocal (v_rcp_f32, 1 ulp) instead of the ~10-instruction
// correctly rounded division: an error of 6e-8 * t in an entry / exit distance is far inside the boxes' padding (2e-5 relative).
MTR_HD float safe_rcp(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    float r = __builtin_amdgcn_rcpf(x);
#else
    float r = 1.0f / x;
#endif
    return (fabsf(r) <= 1e28f) ? r : copysignf(1e28f, x);
}

// ------------------------------------------------------------------------------------------
// Resumable BVH2 traversal.  A traversal is a small per-lane state machine so that a wave can
// interleave node steps, leaf steps and shading of DIFFERENT lanes' rays (the wave-level
// scheduler of k_fused); traverse() below simply runs it to completion.
//   cur >= 0          : at an inner node packet         -> trav_node_step
//   kTravDone < cur<0 : holding a leaf (not yet tested) -> trav_leaf_step
//   cur == kTravDone  : finished, result in h
// Culling is conservative (padded boxes, finite reciprocals): hits are decided by the triangle test alone,
// ties on t by the ORIGINAL triangle index, so the result is independent of the traversal order.
// Stack: reset()/push_if(bool,int)/pop()/empty(); kernels keep it in LDS.
constexpr int32_t kTravDone = (int32_t)0x80000000;
// The far bound of every slab test — min(tmax, closest hit so far) — is widened by 2^-10 (round 6).  A hit distance is what f32
// Moeller-Trumbore computes, and on a grazing sliver that can be off by more than the boxes' padding (measured on the staircase:
// 4e-5 relative): the computed hit then lies in front of its own box, and a walk that has already found a slightly farther hit
// culls the box although the test inside would win — brute force and every tree disagreed on one ray in 1e8 of config 5.  With the
// bound widened a box is culled only when it begins more than 0.1 % behind the best hit; the primitive tests and the tie rule decide
// as before, and the extra boxes visited are those in a shell 0.1 % thick.
constexpr float kCullSlack = 1.0009765625f;

struct Trav {
    f3 o, d, id, noid;
    float tmax;          // culling bound of a node step = min(tmax, h.t): the closest hit so far (h.t stays inf for shadow rays)
    int32_t cur;
    Hit h;
    uint32_t best_orig;
    uint32_t sel[3];      // node_pairs / wide tree: byte offsets of the ENTRY plane pairs of x, y, z inside a node packet (exit = entry ^ 8)
    uint32_t grp;         // wide tree: (node << 9) | (reverse << 8) | mask of the children of `node` still to be visited
};

template <class Stack>
MTR_HD void trav_init(Trav &tr, const SceneView &sc, f3 o, f3 d, float tmax, Stack &st)
{
    tr.o = o; tr.d = d; tr.tmax = tmax;
    // reciprocal direction kept finite so that fma(lo, id, -o*id) never meets inf - inf (axis-parallel rays)
    tr.id = mk(safe_rcp(d.x), safe_rcp(d.y), safe_rcp(d.z));
    tr.noid = mk(-(o.x * tr.id.x), -(o.y * tr.id.y), -(o.z * tr.id.z));
    tr.h.t = kInf; tr.h.u = 0.0f; tr.h.v = 0.0f; tr.h.prim = -1;
    tr.best_orig = 0xffffffffu;
    tr.grp = 0u;
    if (sc.node_pairs || sc.wnodes) {
        const uint32_t sx = tr.id.x < 0.0f ? 8u : 0u, sy = tr.id.y < 0.0f ? 8u : 0u, sz = tr.id.z < 0.0f ? 8u : 0u;
        tr.sel[0] = sx; tr.sel[1] = 16u + sy; tr.sel[2] = 32u + sz;
    }
    tr.cur = sc.n_slots ? 0 : kTravDone;
    st.reset();
}

// one inner-node step: two slab tests, branch-free child selection (unconditional stack write whose
// slot only counts when both children are hit), pop when neither is hit.
template <class Stack>
MTR_HD void trav_node_step(Trav &tr, const SceneView &sc, Stack &st)
{
    st.count(0);
    const f3 id = tr.id, noid = tr.noid;
    const float tb = fminf(tr.tmax, tr.h.t) * kCullSlack;
    // slab planes of both children as packed pairs (.x child 0, .y child 1).  The reciprocal direction is finite
    // (safe_rcp), so fma(p, id, noid) is monotonic in p: the entry plane is `lo` when id >= 0 and `hi` otherwise —
    // selecting it by the sign gives bit for bit what min/max of the two plane distances gives, in fewer instructions.
    f2 nx, fx, ny, fy, nz, fz;
    int32_t c0, c1;
    if (sc.node_pairs) {      // LDS: six 8-byte reads at per-ray offsets (one address computation each) instead of selects
        const char *nb = (const char *)sc.nodes + ((size_t)(uint32_t)tr.cur << 6);
        nx = fma2(*(const f2 *)(nb + tr.sel[0]), id.x, noid.x); fx = fma2(*(const f2 *)(nb + (tr.sel[0] ^ 8u)), id.x, noid.x);
        ny = fma2(*(const f2 *)(nb + tr.sel[1]), id.y, noid.y); fy = fma2(*(const f2 *)(nb + (tr.sel[1] ^ 8u)), id.y, noid.y);
        nz = fma2(*(const f2 *)(nb + tr.sel[2]), id.z, noid.z); fz = fma2(*(const f2 *)(nb + (tr.sel[2] ^ 8u)), id.z, noid.z);
        const f2 cc = *(const f2 *)(nb + 48);
        c0 = (int32_t)fbits(cc.x); c1 = (int32_t)fbits(cc.y);
    } else {
        const Node &n = sc.nodes[tr.cur];
        const q4 X = n.q[0], Y = n.q[1], Z = n.q[2], C = n.q[3];
        const bool sx = id.x < 0.0f, sy = id.y < 0.0f, sz = id.z < 0.0f;
        nx = fma2(sx ? f2{ X.z, X.w } : f2{ X.x, X.y }, id.x, noid.x); fx = fma2(sx ? f2{ X.x, X.y } : f2{ X.z, X.w }, id.x, noid.x);
        ny = fma2(sy ? f2{ Y.z, Y.w } : f2{ Y.x, Y.y }, id.y, noid.y); fy = fma2(sy ? f2{ Y.x, Y.y } : f2{ Y.z, Y.w }, id.y, noid.y);
        nz = fma2(sz ? f2{ Z.z, Z.w } : f2{ Z.x, Z.y }, id.z, noid.z); fz = fma2(sz ? f2{ Z.x, Z.y } : f2{ Z.z, Z.w }, id.z, noid.z);
        c0 = (int32_t)fbits(C.x); c1 = (int32_t)fbits(C.y);
    }
    const float tn0 = fmaxf(fmaxf(nx.x, ny.x), fmaxf(nz.x, 0.0f));
    const float tf0 = fminf(fminf(fx.x, fy.x), fminf(fz.x, tb));
    const float tn1 = fmaxf(fmaxf(nx.y, ny.y), fmaxf(nz.y, 0.0f));
    const float tf1 = fminf(fminf(fx.y, fy.y), fminf(fz.y, tb));
    const bool h0 = tn0 <= tf0, h1 = tn1 <= tf1;
    const bool near0 = tn0 <= tn1;
    const bool both = h0 && h1;
    st.push_if(both, near0 ? c1 : c0);
    int32_t nxt = both ? (near0 ? c0 : c1) : (h0 ? c0 : c1);
    if (!(h0 || h1)) nxt = st.empty() ? kTravDone : st.pop();
    tr.cur = nxt;
}

// one leaf: 1..4 Moller-Trumbore tests, then pop.  `any_hit` is a per-lane runtime flag so that
// closest-hit and shadow rays of different lanes share one instruction stream.
// a rectangle leaf [mitsuba3: Rectangle::ray_intersect_preliminary_impl]: ray to object space (transform_affine: the
// translation first, then one fmadd per column), t = -o.z / d.z, local = fmadd(d, t, o), hit iff 0 <= t <= maxt and
// |local.x|, |local.y| <= 1; (u, v) of the hit = (local.x, local.y)
MTR_HD bool is_quad_leaf(int32_t cur) { return cur < 0 && cur != kTravDone && ((~(uint32_t)cur) & kLeafQuadBit) != 0u; }
template <class Stack>
MTR_HD bool trav_quad_test(Trav &tr, const SceneView &sc, Stack &st, bool any_hit)
{
    st.count(1);
    const uint32_t first = ((~(uint32_t)tr.cur) & ~kLeafQuadBit) >> 2;
    const TriPair &tp = sc.tpairs[first >> 1];
    const q4 Z = tp.g[0], X = tp.g[1], Y = tp.g[2];
    const uint32_t orig = fbits(tp.g[4].z);
    const f3 o = tr.o, d = tr.d;
    const float oz = fmaf(Z.z, o.z, fmaf(Z.y, o.y, fmaf(Z.x, o.x, Z.w))), dz = fmaf(Z.z, d.z, fmaf(Z.y, d.y, Z.x * d.x));
    const float t = -oz / dz;
    const float ox = fmaf(X.z, o.z, fmaf(X.y, o.y, fmaf(X.x, o.x, X.w))), dx = fmaf(X.z, d.z, fmaf(X.y, d.y, X.x * d.x));
    const float oy = fmaf(Y.z, o.z, fmaf(Y.y, o.y, fmaf(Y.x, o.x, Y.w))), dy = fmaf(Y.z, d.z, fmaf(Y.y, d.y, Y.x * d.x));
    const float lx = fmaf(dx, t, ox), ly = fmaf(dy, t, oy);
    const bool hit = (t >= 0.0f) && (t <= tr.tmax) && (fabsf(lx) <= 1.0f) && (fabsf(ly) <= 1.0f);
    const bool closer = (t < tr.h.t) | ((t == tr.h.t) & (orig < tr.best_orig));
    const bool better = hit && (any_hit || closer);
    tr.h.t = better ? t : tr.h.t; tr.h.u = better ? lx : tr.h.u; tr.h.v = better ? ly : tr.h.v;
    tr.h.prim = better ? (int32_t)first : tr.h.prim; tr.best_orig = better ? orig : tr.best_orig;
    return hit;
}

// ONE_PAIR: the caller knows that no triangle leaf of the tree holds more than two triangles (scene trait kTrLeafPair): one
// pass, no loop around it
template <bool ONE_PAIR = false, class Stack>
MTR_HD bool trav_leaf_test(Trav &tr, const SceneView &sc, Stack &st, bool any_hit)
{
    const uint32_t code = ~(uint32_t)tr.cur;
    if (code & kLeafQuadBit) return trav_quad_test(tr, sc, st, any_hit);
    const uint32_t first = code >> 2, cnt = ONE_PAIR ? (code & 1u) + 1u : (code & 3u) + 1u;
    bool found = false;
    // Moller-Trumbore [mitsuba3: Mesh::ray_intersect_triangle]: pvec = cross(d, e2); inv_det = 1 / dot(e1, pvec);
    // tvec = o - p0; u = dot(tvec, pvec) * inv_det; qvec = cross(tvec, e1); v = dot(d, qvec) * inv_det;
    // t = dot(e2, qvec) * inv_det; mitsuba: hit iff 0 <= u <= 1, v >= 0, u + v <= 1, 0 <= t <= tmax  (cross and dot as fma chains).
    // Shared edges are CLOSED (kEdgeEps = 2^-19 of the triangle): the two triangles of an edge evaluate it with different
    // operation orders, so a ray aimed exactly at it can fail both tests by one rounding — which is no measure-zero event
    // when millions of paths connect to ONE point (the laser spot at the centre of a two-triangle relay wall lies on its
    // diagonal: 11 % of the NLOS connections of nlos-z-simple.xml fell through; the reference's Embree evaluates a quad's
    // diagonal with one expression for both triangles).  A ray on the edge hits both at the same t; the tie rule decides.
    // Two triangles of the leaf per pass, one in each half of a register pair (v_pk_mul/add/fma_f32): either half is bit
    // for bit what a scalar evaluation gives; a leaf with an odd count has its last triangle repeated in the pad slot,
    // whose result is ignored.
    for (uint32_t i = 0; i < (ONE_PAIR ? 1u : cnt); i += 2u) {
        st.count(1);
        const bool two = (i + 1u) < cnt;
        const int32_t pa = (int32_t)(first + i), pb = pa + 1;
        const TriPair &tp = sc.tpairs[(first + i) >> 1];
        const q4 g0 = tp.g[0], g1 = tp.g[1], g2 = tp.g[2], g3 = tp.g[3], g4 = tp.g[4];
        const f2 p0x{ g0.x, g0.y }, p0y{ g0.z, g0.w }, p0z{ g1.x, g1.y };
        const f2 e1x{ g1.z, g1.w }, e1y{ g2.x, g2.y }, e1z{ g2.z, g2.w };
        const f2 e2x{ g3.x, g3.y }, e2y{ g3.z, g3.w }, e2z{ g4.x, g4.y };
        const uint32_t orig_a = fbits(g4.z), orig_b = fbits(g4.w);
        const f3 o = tr.o, d = tr.d;
        // pvec = cross(d, e2)
        const f2 pvx = fma2(e2z, d.y, neg2(mul2(e2y, d.z))), pvy = fma2(e2x, d.z, neg2(mul2(e2z, d.x))), pvz = fma2(e2y, d.x, neg2(mul2(e2x, d.y)));
        const f2 det = fma2(e1x, pvx, fma2(e1y, pvy, mul2(e1z, pvz)));                 // dot(e1, pvec)
        const f2 inv_det{ 1.0f / det.x, 1.0f / det.y };
        const f2 tvx = rsub2(o.x, p0x), tvy = rsub2(o.y, p0y), tvz = rsub2(o.z, p0z);  // tvec = o - p0
        const f2 u = mul2(fma2(tvx, pvx, fma2(tvy, pvy, mul2(tvz, pvz))), inv_det);    // dot(tvec, pvec) * inv_det
        // qvec = cross(tvec, e1)
        const f2 qx = fma2(tvy, e1z, neg2(mul2(tvz, e1y))), qy = fma2(tvz, e1x, neg2(mul2(tvx, e1z))), qz = fma2(tvx, e1y, neg2(mul2(tvy, e1x)));
        const f2 v = mul2(fma2(qx, d.x, fma2(qy, d.y, mul2(qz, d.z))), inv_det);       // dot(d, qvec) * inv_det
        const f2 t = mul2(fma2(e2x, qx, fma2(e2y, qy, mul2(e2z, qz))), inv_det);       // dot(e2, qvec) * inv_det
        // hit iff u, v, w = 1 - (u + v) >= -kEdgeEps (mitsuba's `u <= 1` follows from the other three) and 0 <= t <= tmax.  ONE
        // constant: with a second one (1 + eps for `u + v <= 1 + eps`) the scalar registers of k_fused overflowed — 177 more
        // v_readlane in the persistent loop, 65.3 -> 67.9 ms on config 2; this form has the instruction count of the plain test.
        const f2 w = rsub2(1.0f, add2(u, v));
        {
            const bool hit = (u.x >= -kEdgeEps) && (v.x >= -kEdgeEps) && (w.x >= -kEdgeEps) && (t.x >= 0.0f) && (t.x <= tr.tmax);
            const bool closer = (t.x < tr.h.t) | ((t.x == tr.h.t) & (orig_a < tr.best_orig));      // bitwise: no branches for three compares
            const bool better = hit && (any_hit ? !found : closer);
            found = found || hit;
            tr.h.t = better ? t.x : tr.h.t; tr.h.u = better ? u.x : tr.h.u; tr.h.v = better ? v.x : tr.h.v;
            tr.h.prim = better ? pa : tr.h.prim; tr.best_orig = better ? orig_a : tr.best_orig;
        }
        {
            const bool hit = two && (u.y >= -kEdgeEps) && (v.y >= -kEdgeEps) && (w.y >= -kEdgeEps) && (t.y >= 0.0f) && (t.y <= tr.tmax);
            const bool closer = (t.y < tr.h.t) | ((t.y == tr.h.t) & (orig_b < tr.best_orig));
            const bool better = hit && (any_hit ? !found : closer);
            found = found || hit;
            tr.h.t = better ? t.y : tr.h.t; tr.h.u = better ? u.y : tr.h.u; tr.h.v = better ? v.y : tr.h.v;
            tr.h.prim = better ? pb : tr.h.prim; tr.best_orig = better ? orig_b : tr.best_orig;
        }
    }
    return found;
}
template <class Stack>
MTR_HD void trav_leaf_step(Trav &tr, const SceneView &sc, Stack &st, bool any_hit)
{
    const bool found = trav_leaf_test(tr, sc, st, any_hit);
    if (any_hit & found) tr.cur = kTravDone;
    else tr.cur = st.empty() ? kTravDone : st.pop();
}

// BOX nodes (an object that is an affine image of the cube [-1,1]^3, every face split into two triangles: mitsuba's `cube`):
// instead of slab tests against the six faces' boxes, the slab distances of the cube itself give the points pe, px where
// the ray enters and leaves it, and only the faces that can hold the closest hit are visited — normally ONE leaf instead of
// the entry face's and the exit face's.  The selection only prunes; hits, t and barycentrics come from the Moller-Trumbore
// test of the visited leaves, so the result is bit for bit what visiting all six gives as long as the selection is a superset:
//   the faces whose plane pe touches — the entry face, both neighbours at an edge, an exit-side plane the ray grazes or
//   clips a corner through;
//   and, when the entry test may fail although the ray meets the cube — origin inside it; pe on an edge or a corner (two
//   planes touched), or on a diagonal of its face, where the ray can slip between two triangles and meet the far side
//   from within — also the faces px touches.
// "Touches" = within eps of the plane, the point itself within eps of the cube (else the ray misses it).  eps covers the
// rounding of the point several times over: 2e-5 + 8e-6 m per axis (m = the magnitude of the terms of the object-space
// coordinate: its rounding is ~1e-7 m) plus the distance the point slides along that axis when the slab distance is off
// by the rounding of a coordinate divided by a direction component (grazing rays: large, and then simply more faces, up
// to all six, are visited).  That slide per unit of direction is also the uncertainty of the slab distances themselves:
// pe (px) counts only when its distance is neither behind the origin nor beyond tb = min(tmax, closest hit so far) by more
// than it — a ray leaving a face of the cube does not revisit that face.
// Returns the mask of faces, bit f = face 2 * axis + (coordinate = +1).
// box_mag: the magnitude of the terms of the object-space coordinates of a ray's origin, per axis ...
MTR_HD f3 box_mag(q4 A, q4 B, q4 C, f3 o)
{
    return mk(fmaf(fabsf(A.z), fabsf(o.z), fmaf(fabsf(A.y), fabsf(o.y), fmaf(fabsf(A.x), fabsf(o.x), fabsf(A.w)))),
              fmaf(fabsf(B.z), fabsf(o.z), fmaf(fabsf(B.y), fabsf(o.y), fmaf(fabsf(B.x), fabsf(o.x), fabsf(B.w)))),
              fmaf(fabsf(C.z), fabsf(o.z), fmaf(fabsf(C.y), fabsf(o.y), fmaf(fabsf(C.x), fabsf(o.x), fabsf(C.w)))));
}
// ... or an upper bound of it from the row sums S of |A|, |B|, |C| and the largest coordinate of the origin (a larger `mag` only
// widens the tolerances: more faces, never fewer) — three fmas where box_mag takes nine (flat_walk_device)
MTR_HD f3 box_mag_bound(q4 A, q4 B, q4 C, q4 S, float omax)
{
    return mk(fmaf(S.x, omax, fabsf(A.w)), fmaf(S.y, omax, fabsf(B.w)), fmaf(S.z, omax, fabsf(C.w)));
}
MTR_HD uint32_t box_select(f3 mag, f3 ol, f3 dl, f3 id, float tb)
{
    const f3 en = mk(dl.x < 0.0f ? 1.0f : -1.0f, dl.y < 0.0f ? 1.0f : -1.0f, dl.z < 0.0f ? 1.0f : -1.0f);   // entry planes; exit = -en
    const float tn = fmaxf(fmaxf((en.x - ol.x) * id.x, (en.y - ol.y) * id.y), (en.z - ol.z) * id.z);
    const float tf = fminf(fminf((-en.x - ol.x) * id.x, (-en.y - ol.y) * id.y), (-en.z - ol.z) * id.z);
    const float slide = 8e-6f * fmaxf(fmaxf(mag.x * fabsf(id.x), mag.y * fabsf(id.y)), mag.z * fabsf(id.z));
    const f3 eps = mk(fmaf(slide, fabsf(dl.x), fmaf(8e-6f, mag.x, 2e-5f)), fmaf(slide, fabsf(dl.y), fmaf(8e-6f, mag.y, 2e-5f)),
                      fmaf(slide, fabsf(dl.z), fmaf(8e-6f, mag.z, 2e-5f)));
    const f3 pe = mk(fmaf(tn, dl.x, ol.x), fmaf(tn, dl.y, ol.y), fmaf(tn, dl.z, ol.z));
    const f3 px = mk(fmaf(tf, dl.x, ol.x), fmaf(tf, dl.y, ol.y), fmaf(tf, dl.z, ol.z));
    const f3 ae = mk(fabsf(pe.x), fabsf(pe.y), fabsf(pe.z));
    // (`slide` is the uncertainty of tn and tf themselves: a point behind the origin, or beyond the closest hit so far, holds no hit)
    const bool in_e = (ae.x <= 1.0f + eps.x) && (ae.y <= 1.0f + eps.y) && (ae.z <= 1.0f + eps.z) && (tn >= -slide) && (tn <= tb + slide);
    const bool in_x = (fabsf(px.x) <= 1.0f + eps.x) && (fabsf(px.y) <= 1.0f + eps.y) && (fabsf(px.z) <= 1.0f + eps.z) && (tf >= -slide) && (tf <= tb + slide);
    const bool inside = (fabsf(ol.x) <= 1.0f + eps.x) && (fabsf(ol.y) <= 1.0f + eps.y) && (fabsf(ol.z) <= 1.0f + eps.z);
    const uint32_t bx_en = en.x > 0.0f ? 2u : 1u, by_en = en.y > 0.0f ? 8u : 4u, bz_en = en.z > 0.0f ? 32u : 16u;
    uint32_t mask = 0u, later = 0u;
    mask |= (in_e && fabsf(pe.x - en.x) <= eps.x) ? bx_en : 0u;
    mask |= (in_e && fabsf(pe.y - en.y) <= eps.y) ? by_en : 0u;
    mask |= (in_e && fabsf(pe.z - en.z) <= eps.z) ? bz_en : 0u;
    mask |= (in_e && fabsf(pe.x + en.x) <= eps.x) ? (bx_en ^ 3u) : 0u;
    mask |= (in_e && fabsf(pe.y + en.y) <= eps.y) ? (by_en ^ 12u) : 0u;
    mask |= (in_e && fabsf(pe.z + en.z) <= eps.z) ? (bz_en ^ 48u) : 0u;
    later |= (in_x && fabsf(px.x + en.x) <= eps.x) ? (bx_en ^ 3u) : 0u;
    later |= (in_x && fabsf(px.y + en.y) <= eps.y) ? (by_en ^ 12u) : 0u;
    later |= (in_x && fabsf(px.z + en.z) <= eps.z) ? (bz_en ^ 48u) : 0u;
    // a diagonal of the entry face: two of |pe|'s coordinates agree (the third is 1, which the edge test above covers)
    const bool diagonal = (fabsf(ae.x - ae.y) <= eps.x + eps.y) || (fabsf(ae.y - ae.z) <= eps.y + eps.z) || (fabsf(ae.x - ae.z) <= eps.x + eps.z);
    const bool risky = inside || diagonal || (mask & (mask - 1u)) != 0u;
    return mask | (risky ? later : 0u);
}

// ---- wide trees (SceneView::wnodes / wnodes4) ----
// next child of the current group (or of the group on top of the stack): sets tr.cur
template <uint32_t W, class Stack>
MTR_HD void wide_advance(Trav &tr, const void *nodes, Stack &st, uint32_t g)
{
    typedef WNodeT<W> N;
    if ((g & N::kMask) == 0u) g = st.empty() ? 0u : (uint32_t)st.pop();
    const uint32_t mask = g & N::kMask;
    if (mask == 0u) { tr.grp = 0u; tr.cur = kTravDone; return; }
    const uint32_t qm = mask & ((1u << ((g >> N::kQuadShift) & 0xfu)) - 1u);          // rectangle children still to visit
    const uint32_t k = qm ? (uint32_t)__builtin_ctz(qm)
                          : ((g & N::kRevBit) ? 31u - (uint32_t)__builtin_clz(mask) : (uint32_t)__builtin_ctz(mask));
    g &= ~(1u << k);
    tr.grp = g;
    tr.cur = *(const int32_t *)((const char *)nodes + (size_t)(g >> N::kNodeShift) * N::kBytes + N::kRefOff + 4u * k);
}
// one wide-node step: packed pairs of slab tests -> mask of the children the ray enters.
// OFFS: entry / exit planes fetched at sign-dependent byte offsets (LDS) instead of whole quads + selects (HBM).
// (wide_node_test: the step without the advance to the next child — returns the group word to advance from; callers that
// share ONE advance between their node and primitive steps use it directly: wide_walk_device)
template <uint32_t W, bool OFFS, class Stack>
MTR_HD uint32_t wide_node_test(Trav &tr, const void *nodes, Stack &st)
{
    typedef WNodeT<W> N;
    st.count(0);
    f3 id = tr.id, noid = tr.noid;
    const float tb = fminf(tr.tmax, tr.h.t) * kCullSlack;
    const char *nb = (const char *)nodes + (size_t)(uint32_t)tr.cur * N::kBytes;
    const uint32_t axis = *(const uint32_t *)(nb + N::kHdrOff), count = *(const uint32_t *)(nb + N::kHdrOff + 4u);
    const uint32_t n_quads = *(const uint32_t *)(nb + N::kHdrOff + 8u), flags = *(const uint32_t *)(nb + N::kHdrOff + 12u);
    uint32_t sel0 = tr.sel[0], sel1 = tr.sel[1], sel2 = tr.sel[2];
    uint32_t box_m = 0u;
    if (flags & 1u) {          // object node: the ray in the shape's object space (t is invariant under the affine map)
        const q4 A = *(const q4 *)(nb + N::kXfOff), B = *(const q4 *)(nb + N::kXfOff + 16u), C = *(const q4 *)(nb + N::kXfOff + 32u);
        const f3 o = tr.o, d = tr.d;
        const f3 ol = mk(fmaf(A.z, o.z, fmaf(A.y, o.y, fmaf(A.x, o.x, A.w))), fmaf(B.z, o.z, fmaf(B.y, o.y, fmaf(B.x, o.x, B.w))),
                         fmaf(C.z, o.z, fmaf(C.y, o.y, fmaf(C.x, o.x, C.w))));
        const f3 dl = mk(fmaf(A.z, d.z, fmaf(A.y, d.y, A.x * d.x)), fmaf(B.z, d.z, fmaf(B.y, d.y, B.x * d.x)),
                         fmaf(C.z, d.z, fmaf(C.y, d.y, C.x * d.x)));
        id = mk(safe_rcp(dl.x), safe_rcp(dl.y), safe_rcp(dl.z));
        noid = mk(-(ol.x * id.x), -(ol.y * id.y), -(ol.z * id.z));
        sel0 = id.x < 0.0f ? 8u : 0u; sel1 = 16u + (id.y < 0.0f ? 8u : 0u); sel2 = 32u + (id.z < 0.0f ? 8u : 0u);
        if (flags & 2u) box_m = box_select(box_mag(A, B, C, o), ol, dl, id, tb);
    }
    const bool sx = id.x < 0.0f, sy = id.y < 0.0f, sz = id.z < 0.0f;
    uint32_t m = 0u;
    if (flags & 2u) m = box_m;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (uint32_t j = 0; j < W / 2u; ++j) {
        if (OFFS ? (2u * j < ((flags & 2u) ? 0u : count)) : true) {
            const char *pb = nb + 48u * j;
            f2 nx, fx, ny, fy, nz, fz;
            if (OFFS) {
                nx = fma2(*(const f2 *)(pb + sel0), id.x, noid.x); fx = fma2(*(const f2 *)(pb + (sel0 ^ 8u)), id.x, noid.x);
                ny = fma2(*(const f2 *)(pb + sel1), id.y, noid.y); fy = fma2(*(const f2 *)(pb + (sel1 ^ 8u)), id.y, noid.y);
                nz = fma2(*(const f2 *)(pb + sel2), id.z, noid.z); fz = fma2(*(const f2 *)(pb + (sel2 ^ 8u)), id.z, noid.z);
            } else {
                const q4 X = *(const q4 *)pb, Y = *(const q4 *)(pb + 16), Z = *(const q4 *)(pb + 32);
                nx = fma2(sx ? f2{ X.z, X.w } : f2{ X.x, X.y }, id.x, noid.x); fx = fma2(sx ? f2{ X.x, X.y } : f2{ X.z, X.w }, id.x, noid.x);
                ny = fma2(sy ? f2{ Y.z, Y.w } : f2{ Y.x, Y.y }, id.y, noid.y); fy = fma2(sy ? f2{ Y.x, Y.y } : f2{ Y.z, Y.w }, id.y, noid.y);
                nz = fma2(sz ? f2{ Z.z, Z.w } : f2{ Z.x, Z.y }, id.z, noid.z); fz = fma2(sz ? f2{ Z.x, Z.y } : f2{ Z.z, Z.w }, id.z, noid.z);
            }
            const float tn0 = fmaxf(fmaxf(nx.x, ny.x), fmaxf(nz.x, 0.0f));
            const float tf0 = fminf(fminf(fx.x, fy.x), fminf(fz.x, tb));
            const float tn1 = fmaxf(fmaxf(nx.y, ny.y), fmaxf(nz.y, 0.0f));
            const float tf1 = fminf(fminf(fx.y, fy.y), fminf(fz.y, tb));
            m |= (tn0 <= tf0 ? 1u : 0u) << (2u * j);
            m |= (tn1 <= tf1 ? 2u : 0u) << (2u * j);
        }
    }
    uint32_t g = tr.grp;
    if (m != 0u) {
        st.push_if((g & N::kMask) != 0u, (int32_t)g);
        // walk order of this node's children: reversed when the direction is negative on the node's sort axis
        bool neg;
        if (OFFS) neg = ((axis == 0u ? sel0 : (axis == 1u ? sel1 : sel2)) & 8u) != 0u;      // bit 3 = sign (trav_init)
        else neg = axis == 0u ? sx : (axis == 1u ? sy : sz);
        g = ((uint32_t)tr.cur << N::kNodeShift) | (n_quads << N::kQuadShift) | (neg ? N::kRevBit : 0u) | m;
    }
    return g;
}
template <uint32_t W, bool OFFS, class Stack>
MTR_HD void wide_node_step(Trav &tr, const void *nodes, Stack &st)
{
    const uint32_t g = wide_node_test<W, OFFS>(tr, nodes, st);
    wide_advance<W>(tr, nodes, st, g);
}
template <uint32_t W, bool ONE_PAIR = false, class Stack>
MTR_HD void wide_leaf_step(Trav &tr, const SceneView &sc, const void *nodes, Stack &st, bool any_hit)
{
    const bool found = trav_leaf_test<ONE_PAIR>(tr, sc, st, any_hit);
    if (any_hit & found) tr.cur = kTravDone;
    else wide_advance<W>(tr, nodes, st, tr.grp);
}

// ---- quantised 4-wide tree (SceneView::wnodes4) ----
template <class Stack>
MTR_HD void qwide_advance(Trav &tr, const QNode4 *nodes, Stack &st, uint32_t g)
{
    if ((g & 0xfu) == 0u) g = st.empty() ? 0u : (uint32_t)st.pop();
    const uint32_t mask = g & 0xfu;
    if (mask == 0u) { tr.grp = 0u; tr.cur = kTravDone; return; }
    const uint32_t k = (g & 0x10u) ? 31u - (uint32_t)__builtin_clz(mask) : (uint32_t)__builtin_ctz(mask);
    g &= ~(1u << k);
    tr.grp = g;
    tr.cur = *((const int32_t *)&nodes[g >> 5].q[1] + k);
}
MTR_HD float qbyte(uint32_t w, uint32_t c) { return (float)((w >> (8u * c)) & 0xffu); }     // v_cvt_f32_ubyte<c>
template <class Stack>
MTR_HD void qwide_node_step(Trav &tr, const QNode4 *nodes, Stack &st)
{
    st.count(0);
    const QNode4 &n = nodes[tr.cur];
    const q4 A = n.q[0], P = n.q[2], Q = n.q[3];
    const uint32_t meta = fbits(A.w);
    const f3 id = tr.id;
    const float tb = fminf(tr.tmax, tr.h.t) * kCullSlack;
    // plane distance = (org + q * step) * id + noid, evaluated as q * (step * id) + (org * id + noid)
    const float kx = bitsf((meta & 0xffu) << 23) * id.x, ky = bitsf(((meta >> 8) & 0xffu) << 23) * id.y, kz = bitsf(((meta >> 16) & 0xffu) << 23) * id.z;
    const float bx = fmaf(A.x, id.x, tr.noid.x), by = fmaf(A.y, id.y, tr.noid.y), bz = fmaf(A.z, id.z, tr.noid.z);
    const bool sx = id.x < 0.0f, sy = id.y < 0.0f, sz = id.z < 0.0f;
    const uint32_t lox = fbits(P.x), loy = fbits(P.y), loz = fbits(P.z), hix = fbits(P.w), hiy = fbits(Q.x), hiz = fbits(Q.y);
    const uint32_t nxw = sx ? hix : lox, fxw = sx ? lox : hix, nyw = sy ? hiy : loy, fyw = sy ? loy : hiy, nzw = sz ? hiz : loz, fzw = sz ? loz : hiz;
    uint32_t m = 0u;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (uint32_t c = 0; c < 4u; c += 2u) {
        const f2 nx = fma2(f2{ qbyte(nxw, c), qbyte(nxw, c + 1u) }, kx, bx), fx = fma2(f2{ qbyte(fxw, c), qbyte(fxw, c + 1u) }, kx, bx);
        const f2 ny = fma2(f2{ qbyte(nyw, c), qbyte(nyw, c + 1u) }, ky, by), fy = fma2(f2{ qbyte(fyw, c), qbyte(fyw, c + 1u) }, ky, by);
        const f2 nz = fma2(f2{ qbyte(nzw, c), qbyte(nzw, c + 1u) }, kz, bz), fz = fma2(f2{ qbyte(fzw, c), qbyte(fzw, c + 1u) }, kz, bz);
        const float tn0 = fmaxf(fmaxf(nx.x, ny.x), fmaxf(nz.x, 0.0f));
        const float tf0 = fminf(fminf(fx.x, fy.x), fminf(fz.x, tb));
        const float tn1 = fmaxf(fmaxf(nx.y, ny.y), fmaxf(nz.y, 0.0f));
        const float tf1 = fminf(fminf(fx.y, fy.y), fminf(fz.y, tb));
        m |= (tn0 <= tf0 ? 1u : 0u) << c;
        m |= (tn1 <= tf1 ? 2u : 0u) << c;
    }
    m &= (1u << (meta >> 26)) - 1u;                       // absent children
    uint32_t g = tr.grp;
    if (m != 0u) {
        st.push_if((g & 0xfu) != 0u, (int32_t)g);
        const uint32_t axis = (meta >> 24) & 3u;
        const bool neg = axis == 0u ? sx : (axis == 1u ? sy : sz);
        g = ((uint32_t)tr.cur << 5) | (neg ? 0x10u : 0u) | m;
    }
    qwide_advance(tr, nodes, st, g);
}
template <class Stack>
MTR_HD void qwide_leaf_step(Trav &tr, const SceneView &sc, Stack &st, bool any_hit)
{
    const bool found = trav_leaf_test(tr, sc, st, any_hit);
    if (any_hit & found) tr.cur = kTravDone;
    else qwide_advance(tr, sc.wnodes4, st, tr.grp);
}

// ---- quantised 8-wide tree (SceneView::wnodes8q): group word = (node << 9) | (reverse << 8) | mask of children to visit ----
template <class Stack>
MTR_HD void q8_advance(Trav &tr, const QNode8 *nodes, Stack &st, uint32_t g)
{
    if ((g & 0xffu) == 0u) g = st.empty() ? 0u : (uint32_t)st.pop();
    const uint32_t mask = g & 0xffu;
    if (mask == 0u) { tr.grp = 0u; tr.cur = kTravDone; return; }
    const uint32_t k = (g & 0x100u) ? 31u - (uint32_t)__builtin_clz(mask) : (uint32_t)__builtin_ctz(mask);
    g &= ~(1u << k);
    tr.grp = g;
    tr.cur = *((const int32_t *)&nodes[g >> 9].q[4] + k);
}
template <class Stack>
MTR_HD void q8_node_step(Trav &tr, const QNode8 *nodes, Stack &st)
{
    st.count(0);
    const QNode8 &n = nodes[tr.cur];
    const q4 A = n.q[0], P1 = n.q[1], P2 = n.q[2], P3 = n.q[3];
    const uint32_t meta = fbits(A.w);
    const f3 id = tr.id;
    const float tb = fminf(tr.tmax, tr.h.t) * kCullSlack;
    const float kx = bitsf((meta & 0xffu) << 23) * id.x, ky = bitsf(((meta >> 8) & 0xffu) << 23) * id.y, kz = bitsf(((meta >> 16) & 0xffu) << 23) * id.z;
    const float bx = fmaf(A.x, id.x, tr.noid.x), by = fmaf(A.y, id.y, tr.noid.y), bz = fmaf(A.z, id.z, tr.noid.z);
    const bool sx = id.x < 0.0f, sy = id.y < 0.0f, sz = id.z < 0.0f;
    uint32_t m = 0u;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (uint32_t h = 0; h < 2u; ++h) {
        const uint32_t lox = fbits(h ? P1.y : P1.x), loy = fbits(h ? P1.w : P1.z), loz = fbits(h ? P2.y : P2.x);
        const uint32_t hix = fbits(h ? P2.w : P2.z), hiy = fbits(h ? P3.y : P3.x), hiz = fbits(h ? P3.w : P3.z);
        const uint32_t nxw = sx ? hix : lox, fxw = sx ? lox : hix, nyw = sy ? hiy : loy, fyw = sy ? loy : hiy, nzw = sz ? hiz : loz, fzw = sz ? loz : hiz;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (uint32_t c = 0; c < 4u; c += 2u) {
            const f2 nx = fma2(f2{ qbyte(nxw, c), qbyte(nxw, c + 1u) }, kx, bx), fx = fma2(f2{ qbyte(fxw, c), qbyte(fxw, c + 1u) }, kx, bx);
            const f2 ny = fma2(f2{ qbyte(nyw, c), qbyte(nyw, c + 1u) }, ky, by), fy = fma2(f2{ qbyte(fyw, c), qbyte(fyw, c + 1u) }, ky, by);
            const f2 nz = fma2(f2{ qbyte(nzw, c), qbyte(nzw, c + 1u) }, kz, bz), fz = fma2(f2{ qbyte(fzw, c), qbyte(fzw, c + 1u) }, kz, bz);
            const float tn0 = fmaxf(fmaxf(nx.x, ny.x), fmaxf(nz.x, 0.0f));
            const float tf0 = fminf(fminf(fx.x, fy.x), fminf(fz.x, tb));
            const float tn1 = fmaxf(fmaxf(nx.y, ny.y), fmaxf(nz.y, 0.0f));
            const float tf1 = fminf(fminf(fx.y, fy.y), fminf(fz.y, tb));
            m |= (tn0 <= tf0 ? 1u : 0u) << (4u * h + c);
            m |= (tn1 <= tf1 ? 2u : 0u) << (4u * h + c);
        }
    }
    m &= (1u << ((meta >> 26) & 0xfu)) - 1u;                       // absent children
    uint32_t g = tr.grp;
    if (m != 0u) {
        st.push_if((g & 0xffu) != 0u, (int32_t)g);
        const uint32_t axis = (meta >> 24) & 3u;
        const bool neg = axis == 0u ? sx : (axis == 1u ? sy : sz);
        g = ((uint32_t)tr.cur << 9) | (neg ? 0x100u : 0u) | m;
    }
    q8_advance(tr, nodes, st, g);
}
template <class Stack>
MTR_HD void q8_leaf_step(Trav &tr, const SceneView &sc, Stack &st, bool any_hit)
{
    const bool found = trav_leaf_test(tr, sc, st, any_hit);
    if (any_hit & found) tr.cur = kTravDone;
    else q8_advance(tr, sc.wnodes8q, st, tr.grp);
}

#if defined(__HIP_DEVICE_COMPILE__)
// The walk of the 8-wide tree on the device.
//
// PRIMITIVES FIRST: the wave runs an inner-node step only when NONE of its lanes holds a primitive (a lane at a node waits
// for the others' rectangle / triangle tests).  Every lane still executes its own steps in its own order; what changes is
// that the lanes' node steps fall into the same wave iterations: after the root, every lane that enters an object node does
// so in ONE step instead of whenever its own rectangles happen to be done (config 2: node wave-steps 10.9 -> 9.3 M per 64 spp,
// k_fused 73.2 -> 68.5 ms; -DMTR_NODES_FIRST = the former while-while order, which is also what a host build runs).
// Measured and not kept: majority votes between the two phases (68.9 .. 69.8 ms against 68.9), and a COOPERATIVE second
// step — before the wave's second node step, every lane holding a node and a pending sibling handed the sibling's subtree
// to a lane that was already done (ray over ds_bpermute, pairing through the empty LDS stack rows, results merged by the
// (t, original index) rule; all parity tests green): 70.7 ms for closest-hit rays only against 68.2, and 92 ms with the
// shadow rays included (the state alive across the shadow traversal left no registers: 30 -> 118 spilled).
template <bool ANY_HIT, bool ONE_PAIR = false, class Stack>
__device__ __forceinline__ void wide_walk_device(Trav &tr, const SceneView &sc, Stack &st)
{
#ifndef MTR_WALK_LANE_EXIT
#ifdef MTR_PROFILE_CYCLES
    uint32_t n_phase_ = 0u;
#endif
    for (;;) {                          // every lane stays until the whole wave is done: ONE loop exit per wave (config 2: 67.9 ms
                                        // against 68.8 / 70.7 ms for the per-lane exit below, same box)
        const bool act = tr.cur != kTravDone;
        if (__ballot(act) == 0ull) break;
        const bool at_prim = act && tr.cur < 0;
        // ONE advance per wave iteration (round 5): the rectangle test, the triangle-leaf test and the node step only say where to
        // advance from; rounds 2-4 carried three inlined copies of wide_advance, two of which ran one after the other whenever an
        // iteration held rectangle lanes AND triangle-leaf lanes
        uint32_t g = tr.grp;
        bool adv = false;
        if (__ballot(at_prim) != 0ull) {
            if (at_prim) {
                const bool found = is_quad_leaf(tr.cur) ? trav_quad_test(tr, sc, st, ANY_HIT) : trav_leaf_test<ONE_PAIR>(tr, sc, st, ANY_HIT);
                if (ANY_HIT & found) tr.cur = kTravDone;
                else adv = true;
            }
        } else {
#ifdef MTR_PROFILE_CYCLES      // experiment build: section 0 = the walk up to the wave's second node step, 5 = the rest of it
            if (++n_phase_ == 2u) st.prof_mark(0);
#endif
            if (tr.cur >= 0) { g = wide_node_test<kWide, true>(tr, sc.wnodes, st); adv = true; }
        }
        if (adv) wide_advance<kWide>(tr, sc.wnodes, st, g);
    }
#ifdef MTR_PROFILE_CYCLES
    st.prof_mark(n_phase_ >= 2u ? 5 : 0);
#endif
#else
    while (tr.cur != kTravDone) {
        const bool at_prim = tr.cur < 0;
        if (__ballot(at_prim) != 0ull) {
            if (is_quad_leaf(tr.cur)) {
                const bool found = trav_quad_test(tr, sc, st, ANY_HIT);
                if (ANY_HIT & found) tr.cur = kTravDone;
                else wide_advance<kWide>(tr, sc.wnodes, st, tr.grp);
            } else if (at_prim) wide_leaf_step<kWide, ONE_PAIR>(tr, sc, sc.wnodes, st, ANY_HIT);
        } else {
            wide_node_step<kWide, true>(tr, sc.wnodes, st);
        }
    }
#endif
}
#endif

#if defined(__HIPCC__)
// A field of the kernel's (single, by-value) argument read from the kernarg segment with scalar loads AT THE POINT OF USE: the
// empty asm makes the address opaque, so the loads can neither be hoisted out of a persistent loop nor merged with the
// compiler's own copy of the argument.  (Taking the address of the argument itself sends the whole struct through scratch.)
template <class T>
__device__ __forceinline__ T kernarg_copy(size_t offset)
{
    static_assert(sizeof(T) % 4 == 0, "dword-sized fields");
    typedef const uint32_t __attribute__((address_space(4))) *WordPtr;
    WordPtr w = (WordPtr)((const char __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr() + offset);
    asm volatile("" : "+s"(w));
    T out;
    uint32_t *o = (uint32_t *)&out;
#pragma unroll
    for (size_t k = 0; k < sizeof(T) / 4; ++k) o[k] = w[k];
    return out;
}
#endif

#if defined(__HIP_DEVICE_COMPILE__)
// The "walk" of a scene with a FLAT TOP LEVEL (FlatTop).  Nothing here decides a hit: rectangles are intersected by
// trav_quad_test, cube faces by trav_leaf_test, ties go to the original index — the arithmetic of every other walker; what
// is different is only WHICH primitives a ray is spared, and every rule that spares one is one the tree walk applies too:
//   boxes      every lane runs box_select for box b in the SAME iteration, on scalar operands; the faces it returns (a
//              superset of those that can hold the hit, see box_select) of all boxes are then tested in lock step — normally
//              one face per lane that meets a box at all, i.e. one pass;
//   rectangles the slab tests of the root node against the rectangles' padded boxes, with the closest hit so far as the far
//              bound — the root step of the tree walk, taken AFTER the boxes instead of before them — while remembering the
//              nearest entry distance t1 (child k1) and the second nearest t2; the nearest is intersected first, and when
//              the hit so far lies in front of t2 every other candidate is culled exactly as a node step taken now would cull it.
// Shadow rays leave at their first hit; the rectangle tests of a wave whose rays all ended on a box are skipped.
// TOP_LEAVES: the top level may hold triangle leaves beside rectangles and boxes (kTrFlatLeaves).  A compile-time fact, not a test of
// FlatTop::prim_mask at run time: with the general form in the Cornell box's kernel config 2 took 54.1 instead of 52.8 ms (round 6).
template <bool ANY_HIT, bool ONE_PAIR, bool TOP_LEAVES, class Stack>
__device__ __forceinline__ void flat_walk_device(Trav &tr, const SceneView &sc, Stack &st)
{
    typedef WNode N;
    const char *root = (const char *)sc.wnodes;
    const FlatHdr fh = kernarg_copy<FlatHdr>(sc.flat_off);
    bool live = true;
    // ---- boxes: face selection per box, uniform over the wave
    uint32_t fm = 0u;
    const float omax = fmaxf(fmaxf(fabsf(tr.o.x), fabsf(tr.o.y)), fabsf(tr.o.z));
    for (uint32_t b = 0; b < fh.n_boxes; ++b) {
        // (requesting the next box's rows before this one's arithmetic costs 45 scalar moves per iteration: measured, not kept)
        const FlatXf X = kernarg_copy<FlatXf>(sc.flat_off + sizeof(FlatHdr) + sizeof(FlatXf) * b);
        const q4 A = X.A, B = X.B, C = X.C, S = X.S;
        const f3 o = tr.o, d = tr.d;
        const f3 ol = mk(fmaf(A.z, o.z, fmaf(A.y, o.y, fmaf(A.x, o.x, A.w))), fmaf(B.z, o.z, fmaf(B.y, o.y, fmaf(B.x, o.x, B.w))),
                         fmaf(C.z, o.z, fmaf(C.y, o.y, fmaf(C.x, o.x, C.w))));
        const f3 dl = mk(fmaf(A.z, d.z, fmaf(A.y, d.y, A.x * d.x)), fmaf(B.z, d.z, fmaf(B.y, d.y, B.x * d.x)),
                         fmaf(C.z, d.z, fmaf(C.y, d.y, C.x * d.x)));
        const f3 id = mk(safe_rcp(dl.x), safe_rcp(dl.y), safe_rcp(dl.z));
        fm |= box_select(box_mag_bound(A, B, C, S, omax), ol, dl, id, tr.tmax) << (6u * b);
    }
    st.prof_flat(0);
    // ---- the selected faces, in lock step (box = k / 6, face = k % 6; the face's leaf = child `face` of the box node)
    while (__ballot(live && fm != 0u) != 0ull) {
        if (live && fm != 0u) {
            const uint32_t k = (uint32_t)__builtin_ctz(fm);
            fm &= fm - 1u;
            const uint32_t box = (k * 43u) >> 8, face = k - 6u * box;
            tr.cur = *(const int32_t *)(root + (size_t)(fh.node0 + box) * N::kBytes + N::kRefOff + 4u * face);
            const bool found = trav_leaf_test<true>(tr, sc, st, ANY_HIT);
            if (ANY_HIT && found) live = false;
        }
    }
    st.prof_flat(5);
    // ---- rectangles
    // (a segment-box pre-test for shadow rays — the box of [o, o + d tmax] against the rectangles' boxes, 12 comparisons per pair
    // instead of the slab tests — was measured in round 6: 53.4 -> 53.9 ms; some lane of nearly every wave grazes the light's box)
    if (__ballot(live) != 0ull) {
        const f3 id = tr.id, noid = tr.noid;
        const float tb = fminf(tr.tmax, tr.h.t);          // (no kCullSlack: rectangles and cube faces are no slivers — and the two multiplies cost config 2 1.6 %)
        const uint32_t sel0 = tr.sel[0], sel1 = tr.sel[1], sel2 = tr.sel[2];
        uint32_t qm = 0u;
        // nearest / second nearest entry distance as ORDERED KEYS: the bits of a non-negative float order like the float, so
        // key = (bits(tn) & ~7) | child orders the candidates by entry distance with the child index along for the ride — one
        // v_min_u32 / v_max_u32 pair per child instead of compares and selects on (distance, index).  Clearing the low bits moves a
        // distance DOWN by at most 7 ulps: the comparison with the hit distance below only gets more careful.
        uint32_t key1 = 0xffffffffu, key2 = 0xffffffffu;
        // one pair of children of the root per step; the offsets are compile-time constants (the loop is unrolled: a pair whose
        // children are absent has inverted boxes and fails by itself), a shadow ray keeps no entry distances
        auto slab_pair = [&](uint32_t j) {
            const char *pb = root + 48u * j;
            const f2 nx = fma2(*(const f2 *)(pb + sel0), id.x, noid.x), fx = fma2(*(const f2 *)(pb + (sel0 ^ 8u)), id.x, noid.x);
            const f2 ny = fma2(*(const f2 *)(pb + sel1), id.y, noid.y), fy = fma2(*(const f2 *)(pb + (sel1 ^ 8u)), id.y, noid.y);
            const f2 nz = fma2(*(const f2 *)(pb + sel2), id.z, noid.z), fz = fma2(*(const f2 *)(pb + (sel2 ^ 8u)), id.z, noid.z);
            const float tn0 = fmaxf(fmaxf(nx.x, ny.x), fmaxf(nz.x, 0.0f));
            const float tf0 = fminf(fminf(fx.x, fy.x), fminf(fz.x, tb));
            const float tn1 = fmaxf(fmaxf(nx.y, ny.y), fmaxf(nz.y, 0.0f));
            const float tf1 = fminf(fminf(fx.y, fy.y), fminf(fz.y, tb));
            const bool h0 = (tn0 <= tf0) && (TOP_LEAVES ? ((fh.prim_mask >> (2u * j)) & 1u) != 0u : 2u * j < fh.n_quads);
            const bool h1 = (tn1 <= tf1) && (TOP_LEAVES ? ((fh.prim_mask >> (2u * j + 1u)) & 1u) != 0u : 2u * j + 1u < fh.n_quads);
            qm |= (h0 ? 1u : 0u) << (2u * j);
            qm |= (h1 ? 2u : 0u) << (2u * j);
            if (!ANY_HIT) {
                const uint32_t ka = h0 ? ((fbits(tn0) & ~7u) | (2u * j)) : 0xffffffffu, kb = h1 ? ((fbits(tn1) & ~7u) | (2u * j + 1u)) : 0xffffffffu;
                key2 = min(key2, max(key1, ka)); key1 = min(key1, ka);
                key2 = min(key2, max(key1, kb)); key1 = min(key1, kb);
            }
        };
        if (TOP_LEAVES) {
            slab_pair(0u);
            if (fh.prim_mask > 3u) slab_pair(1u);
            if (fh.prim_mask > 15u) slab_pair(2u);
            if (kWide > 6u && fh.prim_mask > 63u) slab_pair(3u);
        } else {
            slab_pair(0u); slab_pair(1u); slab_pair(2u);
            if (kWide > 6u && fh.n_quads > 6u) slab_pair(3u);
        }
        if (!live) qm = 0u;
        st.prof_flat(2);
        for (uint32_t it = 0; __ballot(qm != 0u) != 0ull; ++it) {
            if (qm != 0u) {
                const uint32_t k = (!ANY_HIT && it == 0u) ? (key1 & 7u) : (uint32_t)__builtin_ctz(qm);
                qm &= ~(1u << k);
                tr.cur = *(const int32_t *)(root + N::kRefOff + 4u * k);
                // (a rectangle or — only where the top level holds triangle leaves — a leaf of one or two pairs)
                const bool found = (!TOP_LEAVES || is_quad_leaf(tr.cur)) ? trav_quad_test(tr, sc, st, ANY_HIT) : trav_leaf_test<ONE_PAIR>(tr, sc, st, ANY_HIT);
                if (ANY_HIT ? found : (it == 0u && tr.h.t < bitsf(key2 & ~7u))) qm = 0u;
            }
        }
    }
    st.prof_flat(4);
    tr.cur = kTravDone;
}
#endif

// run-to-completion ("while-while": the wave walks inner nodes until every lane holds a leaf, then
// intersects leaves together)
// FLAT: the scene has a flat top level (kTrFlatTop, FlatTop; 2: one that may hold triangle leaves, kTrFlatLeaves) — the device does not walk its
// tree at all (flat_walk_device); a host build walks the tree, whose result is the same by construction
template <bool ANY_HIT, bool ONE_PAIR = false, int FLAT = 0, class Stack>
MTR_HD Hit traverse(const SceneView &sc, f3 o, f3 d, float tmax, Stack &st)
{
    Trav tr;
    trav_init(tr, sc, o, d, tmax, st);
#ifdef MTR_PROFILE_SIMT
    uint32_t my_nodes = 0;
#endif
#if defined(__HIP_DEVICE_COMPILE__)
    if (FLAT) {
        if (tr.cur != kTravDone) flat_walk_device<ANY_HIT, ONE_PAIR, FLAT == 2>(tr, sc, st);
        return tr.h;
    }
#endif
    if (sc.wnodes) {
        // three kinds of steps, each run by the whole wave at once: inner nodes until no lane holds one, then one
        // rectangle test for the lanes holding a rectangle, else one triangle-leaf pass
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MTR_NODES_FIRST)
        wide_walk_device<ANY_HIT, ONE_PAIR>(tr, sc, st);
#else
        while (tr.cur != kTravDone) {
            while (tr.cur >= 0) {
                wide_node_step<kWide, true>(tr, sc.wnodes, st);
#ifdef MTR_PROFILE_SIMT
                ++my_nodes;
#endif
            }
            if (is_quad_leaf(tr.cur)) {
                const bool found = trav_quad_test(tr, sc, st, ANY_HIT);
                if (ANY_HIT & found) tr.cur = kTravDone;
                else wide_advance<kWide>(tr, sc.wnodes, st, tr.grp);
            } else if (tr.cur != kTravDone) wide_leaf_step<kWide, ONE_PAIR>(tr, sc, sc.wnodes, st, ANY_HIT);
        }
#endif
    } else if (sc.wnodes8q) {
        while (tr.cur != kTravDone) {
            while (tr.cur >= 0) q8_node_step(tr, sc.wnodes8q, st);
            if (tr.cur != kTravDone) q8_leaf_step(tr, sc, st, ANY_HIT);
        }
    } else if (sc.wnodes4) {
        while (tr.cur != kTravDone) {
            while (tr.cur >= 0) qwide_node_step(tr, sc.wnodes4, st);
            if (tr.cur != kTravDone) qwide_leaf_step(tr, sc, st, ANY_HIT);
        }
    } else
    while (tr.cur != kTravDone) {
        while (tr.cur >= 0) {
            trav_node_step(tr, sc, st);
#ifdef MTR_PROFILE_SIMT
            ++my_nodes;
#endif
        }
        if (tr.cur != kTravDone) trav_leaf_step(tr, sc, st, ANY_HIT);
    }
#ifdef MTR_PROFILE_SIMT
    st.tail(my_nodes);
#endif
    return tr.h;
}

// ---------------------------------------------------------------- sensor
MTR_HD Ray camera_ray(const Camera &c, const RenderConst &rc, uint32_t px, uint32_t py, float j1, float j2)
{
    float sx = fmaf((float)px + j1, rc.inv_crop_w, rc.off_x);
    float sy = fmaf((float)py + j2, rc.inv_crop_h, rc.off_y);
    const float *M = c.s2c;
    float nx = fmaf(M[0], sx, fmaf(M[1], sy, M[3]));
    float ny = fmaf(M[4], sx, fmaf(M[5], sy, M[7]));
    float nz = fmaf(M[8], sx, fmaf(M[9], sy, M[11]));
    float nw = fmaf(M[12], sx, fmaf(M[13], sy, M[15]));
    float iw = 1.0f / nw;
    f3 dl = normalize(mk(nx * iw, ny * iw, nz * iw));
    const float *T = c.tw;
    Ray r;
    r.d = mk(fmaf(T[0], dl.x, fmaf(T[1], dl.y, T[2] * dl.z)),
             fmaf(T[4], dl.x, fmaf(T[5], dl.y, T[6] * dl.z)),
             fmaf(T[8], dl.x, fmaf(T[9], dl.y, T[10] * dl.z)));
    float inv_z = 1.0f / dl.z;
    float near_t = c.near_clip * inv_z, far_t = c.far_clip * inv_z;
    r.o = fma3(r.d, near_t, mk(T[3], T[7], T[11]));
    r.tmax = far_t - near_t;
    return r;
}

// ---------------------------------------------------------------- film
// returns the time bin or -1 (transient_hdr_film.py:263-265)
MTR_HD int32_t film_bin(const Film &f, float opl)
{
    if (f.n_freq) return (fabsf(opl) <= 3.402823466e+38f) ? 0 : -1;      // phasor film: active &= isfinite(opl) (phasor_image_block.py:47)
    float pos = (opl - f.start_opl) / f.bin_width;
    if (!(pos >= 0.0f && pos < (float)f.tbins)) return -1;
    return (int32_t)(uint32_t)floorf(pos);
}

// row position of a contribution of an exhaustive_scan film: [laser][t]; returns -1 when either is out of range
MTR_HD int32_t film_row_bin(const Film &f, float opl, uint32_t laser)
{
    const int32_t b = film_bin(f, opl);
    return (b < 0 || laser >= f.lasers) ? -1 : (int32_t)(laser * f.tbins + (uint32_t)b);
}

// ---------------------------------------------------------------- BSDFs
MTR_HD float fresnel_conductor(float ci, float er, float ei)
{
    float c2 = ci * ci, s2 = 1.0f - c2, s4 = s2 * s2;
    float t1 = er * er - ei * ei - s2;
    float q = t1 * t1 + 4.0f * ei * ei * er * er;
    float a2pb2 = sqrtf(q > 0.0f ? q : 0.0f);
    float hh = 0.5f * (a2pb2 + t1);
    float a = sqrtf(hh > 0.0f ? hh : 0.0f);
    float term1 = a2pb2 + c2, term2 = 2.0f * ci * a;
    float rs = (term1 - term2) / (term1 + term2);
    float term3 = a2pb2 * c2 + s4, term4 = term2 * s2;
    float rp = rs * (term3 - term4) / (term3 + term4);
    return 0.5f * (rs + rp);
}
MTR_HD void fresnel_dielectric(float ci, float eta, float &r, float &cos_t, float &eta_it, float &eta_ti)
{
    bool outside = ci >= 0.0f;
    float rcp_eta = 1.0f / eta;
    eta_it = outside ? eta : rcp_eta;
    eta_ti = outside ? rcp_eta : eta;
    float ct2 = fmaf(-fmaf(-ci, ci, 1.0f), eta_ti * eta_ti, 1.0f);
    float cia = fabsf(ci), cta = sqrtf(ct2 > 0.0f ? ct2 : 0.0f);
    bool matched = eta == 1.0f, special = matched || cia == 0.0f;
    float as = fmaf(-eta_it, cta, cia) / fmaf(eta_it, cta, cia);
    float ap = fmaf(-eta_it, cia, cta) / fmaf(eta_it, cia, cta);
    float rr = 0.5f * (as * as + ap * ap);
    if (special) rr = matched ? 0.0f : 1.0f;
    r = rr;
    cos_t = sign_neg(ci) ? cta : -cta;
}

// ---- GGX microfacet lobes (MTR_BSDF_ROUGHCONDUCTOR, MTR_BSDF_ROUGHPLASTIC) ----
// Restated from mitsuba 3's MicrofacetDistribution (isotropic alpha, sample_visible = true), RoughConductor and
// RoughPlastic; operation order is the numerics contract shared with the test oracle (fma only where written).
// [mitsuba3: MicrofacetDistribution::eval] D(m) = 1 / (pi alpha^2 ((m.x/alpha)^2 + (m.y/alpha)^2 + m.z^2)^2), 0 when D cos <= 1e-20
MTR_HD float ggx_eval(f3 m, float au, float av)
{
    const float mx = m.x / au, my = m.y / av;
    const float t = fmaf(m.z, m.z, fmaf(my, my, mx * mx));
    const float result = 1.0f / (((kPi * (au * av)) * t) * t);
    return (result * m.z > 1e-20f) ? result : 0.0f;
}
// [MicrofacetDistribution::smith_g1] 2 / (1 + sqrt(1 + alpha^2 tan^2)); 1 at perpendicular incidence; 0 when v sees the back of m
MTR_HD float ggx_smith_g1(f3 v, f3 m, float au, float av)
{
    const float ax = au * v.x, ay = av * v.y;
    const float xy_alpha_2 = fmaf(ay, ay, ax * ax);
    const float tan_theta_alpha_2 = xy_alpha_2 / (v.z * v.z);
    float result = 2.0f / (1.0f + sqrtf(1.0f + tan_theta_alpha_2));
    if (xy_alpha_2 == 0.0f) result = 1.0f;
    if (dot(v, m) * v.z <= 0.0f) result = 0.0f;
    return result;
}
// [MicrofacetDistribution::sample_visible_11] slope of a visible normal for incidence cos_theta_i, unit roughness
MTR_HD void ggx_sample_visible_11(float cos_theta_i, float u1, float u2, float &sx, float &sy)
{
    float px, py;
    concentric_disk(u1, u2, px, py);
    const float s = 0.5f * (1.0f + cos_theta_i);
    const float a0 = fmaf(-px, px, 1.0f);
    const float a = sqrtf(a0 > 0.0f ? a0 : 0.0f);
    py = fmaf(py, s, fmaf(-a, s, a));                        // lerp(a, py, s)
    const float z0 = 1.0f - fmaf(py, py, px * px);
    const float z = sqrtf(z0 > 0.0f ? z0 : 0.0f);
    const float si0 = fmaf(-cos_theta_i, cos_theta_i, 1.0f);
    const float sin_theta_i = sqrtf(si0 > 0.0f ? si0 : 0.0f);
    const float norm = 1.0f / fmaf(sin_theta_i, py, cos_theta_i * z);
    sx = fmaf(cos_theta_i, py, -(sin_theta_i * z)) * norm;
    sy = px * norm;
}
// ---- Beckmann lobes (MTR_MAT_BECKMANN; mitsuba's DEFAULT `distribution`) ----
// exp / log / erf / erfinv restated with explicit operation order — the SAME sequences in the oracle (mtr_oracle.c) — because
// libm, ocml and drjit each round these differently and the numerics contract wants oracle and kernels bit for bit:
//   mtr_expf, mtr_logf: Cephes-style range reduction + polynomial (max. relative error 8e-8 against f64 on [-87, 88] / normal floats);
//   mtr_erff: Abramowitz & Stegun 7.1.28, 1 - (1 + a1 x + .. + a6 x^6)^-16 (max. absolute error 1.8e-6 in f32; erf(inf) = 1);
//   mtr_erfinvf: M. Giles, "Approximating the erfinv function" (2010), single precision (max. relative error 2.6e-7).
// (tests/test_rough_bsdf.py::test_special_functions holds them to scipy.)
MTR_HD float mtr_expf(float x)
{
    if (!(x > -87.0f)) return 0.0f;
    if (x > 88.0f) x = 88.0f;
    const float n = floorf(fmaf(x, 1.44269504088896341f, 0.5f));
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    const float y = fmaf(p, r * r, r) + 1.0f;
    return y * bitsf((uint32_t)((int32_t)n + 127) << 23);
}
MTR_HD float mtr_logf(float x)          // x: a positive normal float
{
    const uint32_t b = fbits(x);
    int32_t e = (int32_t)(b >> 23) - 127;
    float m = bitsf((b & 0x007fffffu) | 0x3f800000u);
    if (m > 1.41421356237f) { m *= 0.5f; e += 1; }
    const float f = m - 1.0f, z = f * f, fe = (float)e;
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
MTR_HD float mtr_erff(float x)
{
    const float a = fabsf(x);
    float t = 0.0000430638f;
    t = fmaf(t, a, 0.0002765672f);
    t = fmaf(t, a, 0.0001520143f);
    t = fmaf(t, a, 0.0092705272f);
    t = fmaf(t, a, 0.0422820123f);
    t = fmaf(t, a, 0.0705230784f);
    t = fmaf(t, a, 1.0f);
    t = t * t; t = t * t; t = t * t; t = t * t;
    const float r = 1.0f - 1.0f / t;
    return x < 0.0f ? -r : r;
}
MTR_HD float mtr_erfinvf(float x)       // |x| < 1
{
    float w = -mtr_logf((1.0f - x) * (1.0f + x));
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
// [MicrofacetDistribution::eval, Beckmann] D(m) = exp(-((m.x/alpha)^2 + (m.y/alpha)^2) / cos^2) / (pi alpha^2 cos^4), 0 when D cos <= 1e-20
MTR_HD float beck_eval(f3 m, float au, float av)
{
    const float mx = m.x / au, my = m.y / av;
    const float c2 = m.z * m.z;
    const float result = mtr_expf(-fmaf(my, my, mx * mx) / c2) / ((kPi * (au * av)) * (c2 * c2));
    return (result * m.z > 1e-20f) ? result : 0.0f;
}
// [MicrofacetDistribution::smith_g1, Beckmann] the rational approximation of Walter et al. in a = 1 / (alpha tan theta)
MTR_HD float beck_smith_g1(f3 v, f3 m, float au, float av)
{
    const float ax = au * v.x, ay = av * v.y;
    const float xy_alpha_2 = fmaf(ay, ay, ax * ax);
    const float tan_theta_alpha_2 = xy_alpha_2 / (v.z * v.z);
    const float a = 1.0f / sqrtf(tan_theta_alpha_2), a_sqr = a * a;
    float result = (a >= 1.6f) ? 1.0f : fmaf(2.181f, a_sqr, 3.535f * a) / fmaf(2.577f, a_sqr, fmaf(2.276f, a, 1.0f));
    if (xy_alpha_2 == 0.0f) result = 1.0f;
    if (dot(v, m) * v.z <= 0.0f) result = 0.0f;
    return result;
}
// [MicrofacetDistribution::sample_visible_11, Beckmann] numerical inversion of the visible-slope CDF in the erf domain:
// a closed-form first guess and three Newton iterations, then the slope across from the second number.  (x is kept
// inside (-1, 1) before every erfinv: a Newton step that overshoots must not produce a NaN slope.)
MTR_HD void beck_sample_visible_11(float cos_theta_i, float u1, float u2, float &sx, float &sy)
{
    const float kInvSqrtPi = 0.56418958354775628695f, kEdge = 0.999999f;
    const float s0 = fmaf(-cos_theta_i, cos_theta_i, 1.0f);
    const float tan_theta_i = sqrtf(s0 > 0.0f ? s0 : 0.0f) / cos_theta_i;
    const float cot_theta_i = 1.0f / tan_theta_i;
    const float maxval = mtr_erff(cot_theta_i);
    u1 = fmaxf(fminf(u1, 1.0f - 1e-6f), 1e-6f); u2 = fmaxf(fminf(u2, 1.0f - 1e-6f), 1e-6f);
    float x = maxval - (maxval + 1.0f) * mtr_erff(sqrtf(-mtr_logf(u1)));
    const float target = u1 * ((1.0f + maxval) + (kInvSqrtPi * tan_theta_i) * mtr_expf(-(cot_theta_i * cot_theta_i)));
    for (int it = 0; it < 3; ++it) {
        x = fmaxf(fminf(x, kEdge), -kEdge);
        const float slope = mtr_erfinvf(x);
        const float value = fmaf(kInvSqrtPi * tan_theta_i, mtr_expf(-(slope * slope)), 1.0f + x) - target;
        const float derivative = 1.0f - slope * tan_theta_i;
        x -= value / derivative;
    }
    x = fmaxf(fminf(x, kEdge), -kEdge);
    sx = mtr_erfinvf(x);
    sy = mtr_erfinvf(fmaf(2.0f, u2, -1.0f));
}
// the two distributions behind one switch (MTR_MAT_BECKMANN in mtr_material.flags)
// (au, av: roughness along the tangent / the bitangent of the shading frame; equal unless MTR_MAT_ANISOTROPIC)
MTR_HD float mf_eval(f3 m, float au, float av, bool beck) { return beck ? beck_eval(m, au, av) : ggx_eval(m, au, av); }
MTR_HD float mf_smith_g1(f3 v, f3 m, float au, float av, bool beck) { return beck ? beck_smith_g1(v, m, au, av) : ggx_smith_g1(v, m, au, av); }
// [RoughConductor: alpha_u, alpha_v] the second roughness of an anisotropic roughconductor travels in c2[0] (a field conductors do not use)
// (roughdielectric: in b[0] — its c2 is the specular transmittance)
MTR_HD float rough_alpha_v(const mtr_material &m)
{
    return (m.flags & MTR_MAT_ANISOTROPIC) ? (m.type == MTR_BSDF_ROUGHDIELECTRIC ? m.b[0] : m.c2[0]) : m.alpha;
}

// [MicrofacetDistribution::sample, sample_visible] visible normal for wi (cos_theta(wi) > 0) and its density
MTR_HD f3 ggx_sample(f3 wi, float au, float av, float u1, float u2, float &pdf, bool beck = false)
{
    const f3 wi_p = normalize(mk(au * wi.x, av * wi.y, wi.z));                  // 1: stretch
    const float sin_theta_2 = fmaf(-wi_p.z, wi_p.z, 1.0f);
    float sin_phi = 0.0f, cos_phi = 1.0f;                                        // Frame3f::sincos_phi
    if (fabsf(sin_theta_2) > 4.0f * 5.9604644775390625e-8f) {
        const float inv = 1.0f / sqrtf(sin_theta_2);
        sin_phi = fminf(fmaxf(wi_p.y * inv, -1.0f), 1.0f); cos_phi = fminf(fmaxf(wi_p.x * inv, -1.0f), 1.0f);
    }
    float sx, sy;
    if (beck) beck_sample_visible_11(wi_p.z, u1, u2, sx, sy);                    // 2: P22 of the stretched direction
    else ggx_sample_visible_11(wi_p.z, u1, u2, sx, sy);
    const float rx = fmaf(cos_phi, sx, -(sin_phi * sy)) * au;                    // 3: rotate, unstretch
    const float ry = fmaf(sin_phi, sx, cos_phi * sy) * av;
    const f3 m = normalize(mk(-rx, -ry, 1.0f));                                  // 4: normal
    pdf = ((mf_eval(m, au, av, beck) * mf_smith_g1(wi, m, au, av, beck)) * fabsf(dot(wi, m))) / wi.z;
    return m;
}
// [RoughPlastic: lerp_gather(m_external_transmittance, cos_theta, MI_ROUGH_TRANSMITTANCE_RES)]
MTR_HD float rough_transmittance(const mtr_material &m, float cos_theta)
{
    const float x = cos_theta * (float)(MTR_ROUGH_TRANSMITTANCE_RES - 1);
    uint32_t i = (uint32_t)x;
    if (i > MTR_ROUGH_TRANSMITTANCE_RES - 2u) i = MTR_ROUGH_TRANSMITTANCE_RES - 2u;
    const float w1 = x - (float)i, w0 = 1.0f - w1;
    return fmaf(w0, m.external_transmittance[i], w1 * m.external_transmittance[i + 1u]);
}
// the SMOOTH lobes of the extended shading code (emitter sampling + MIS apply; eval / pdf / sample through rough_eval_pdf and
// rough_sample): the microfacet lobes and — round 4 — `plastic`, whose diffuse base is smooth and whose coat is a delta lobe
MTR_HD bool bsdf_is_rough(uint32_t type)
{
    return type == MTR_BSDF_ROUGHCONDUCTOR || type == MTR_BSDF_ROUGHPLASTIC || type == MTR_BSDF_ROUGHDIELECTRIC || type == MTR_BSDF_PLASTIC;
}
// ---- plastic (MTR_BSDF_PLASTIC; round 4) [mitsuba3: src/bsdfs/plastic.cpp] ----
// a smooth dielectric coat over a diffuse base with internal scattering: eval / pdf see the diffuse lobe only (the coat is a
// delta lobe), internal_reflectance = fresnel_diffuse_reflectance(1 / eta) (computed by the caller, as mitsuba does at build time)
MTR_HD void plastic_probs(const mtr_material &m, float f_i, float &ps, float &pdif)
{
    ps = f_i * m.specular_sampling_weight; pdif = (1.0f - f_i) * (1.0f - m.specular_sampling_weight);
    ps = ps / (ps + pdif); pdif = 1.0f - ps;
}
MTR_HD void plastic_eval_pdf(const mtr_material &m, f3 albedo, f3 wi, f3 wo, f3 &val, float &pdf)
{
    val = mk(0, 0, 0); pdf = 0.0f;
    const float ci = wi.z, co = wo.z;
    if (!(ci > 0.0f && co > 0.0f)) return;
    const float eta = m.int_ior / m.ext_ior, inv_eta_2 = 1.0f / (eta * eta);
    float f_i, f_o, ct, eit, eti;
    fresnel_dielectric(ci, eta, f_i, ct, eit, eti);
    fresnel_dielectric(co, eta, f_o, ct, eit, eti);
    float ps, pdif;
    plastic_probs(m, f_i, ps, pdif);
    const float cpdf = kInvPi * co;
    pdf = cpdf * pdif;
    const float scale = ((cpdf * inv_eta_2) * (1.0f - f_i)) * (1.0f - f_o);
    const float a[3] = { albedo.x, albedo.y, albedo.z };
    float o[3];
    for (int k = 0; k < 3; ++k)
        o[k] = (a[k] / (1.0f - ((m.flags & MTR_MAT_NONLINEAR) ? a[k] * m.internal_reflectance : m.internal_reflectance))) * scale;
    val = mk(o[0], o[1], o[2]);
}

// ---- rough dielectric interface (MTR_BSDF_ROUGHDIELECTRIC; round 4) ----
// [mitsuba3: src/bsdfs/roughdielectric.cpp, sample_visible = true, TransportMode::Radiance] Walter et al.'s microfacet model of a
// refractive interface: reflection and transmission lobes around the half-vector / the generalised half-vector
// m = normalize(wi + wo * eta).  wi may come from either side (no two-sided wrapper: the plugin is transmissive).
MTR_HD f3 mulsign3(f3 v, float s) { return sign_neg(s) ? mk(-v.x, -v.y, -v.z) : v; }
MTR_HD void rough_dielectric_eval_pdf(const mtr_material &m, f3 wi, f3 wo, f3 &val, float &pdf)
{
    val = mk(0, 0, 0); pdf = 0.0f;
    const float ci = wi.z, co = wo.z;
    if (ci == 0.0f) return;
    const bool beck = (m.flags & MTR_MAT_BECKMANN) != 0u;
    const float au = m.alpha, av = rough_alpha_v(m);
    const bool reflect = ci * co > 0.0f;
    const float eta_m = m.int_ior / m.ext_ior, inv_eta_m = m.ext_ior / m.int_ior;
    const float eta = ci > 0.0f ? eta_m : inv_eta_m, inv_eta = ci > 0.0f ? inv_eta_m : eta_m;
    const float sc = reflect ? 1.0f : eta;
    f3 h = normalize(mk(fmaf(wo.x, sc, wi.x), fmaf(wo.y, sc, wi.y), fmaf(wo.z, sc, wi.z)));
    h = mulsign3(h, h.z);                                             // into the hemisphere of the macro normal
    const float D = mf_eval(h, au, av, beck);
    const float wih = dot(wi, h), woh = dot(wo, h);
    float F, ct, eit, eti;
    fresnel_dielectric(wih, eta_m, F, ct, eit, eti);
    const float g1i = mf_smith_g1(wi, h, au, av, beck);
    const float G = g1i * mf_smith_g1(wo, h, au, av, beck);
    const float t = fmaf(eta, woh, wih);
    if (reflect) {
        const float v = ((F * D) * G) / (4.0f * fabsf(ci));
        val = mk(m.c[0] * v, m.c[1] * v, m.c[2] * v);
    } else {
        // radiance is scaled by the solid-angle compression across the interface (1 / eta^2)
        const float scale = inv_eta * inv_eta;
        const float v = fabsf(((((((scale * (1.0f - F)) * D) * G) * eta) * eta) * wih) * woh / (ci * (t * t)));
        val = mk(m.c2[0] * v, m.c2[1] * v, m.c2[2] * v);
    }
    // the sides micro- and macro-surface must agree on (what smith_g1 enforces in eval and sample)
    if (!(wih * ci > 0.0f && woh * co > 0.0f)) return;
    const float dwh_dwo = reflect ? 1.0f / (4.0f * woh) : ((eta * eta) * woh) / (t * t);
    const f3 wu = mulsign3(wi, ci);
    const float pm = ((D * mf_smith_g1(wu, h, au, av, beck)) * fabsf(dot(wu, h))) / wu.z;       // MicrofacetDistribution::pdf, visible normals
    pdf = fabsf((pm * dwh_dwo) * (reflect ? F : 1.0f - F));
}
// value (cosine included) and density of a rough lobe for local directions; wi, wo already on the two-sided side
// [RoughConductor::eval / ::pdf, RoughPlastic::eval / ::pdf]
MTR_HD void rough_eval_pdf(const mtr_material &m, f3 albedo, f3 wi, f3 wo, f3 &val, float &pdf)
{
    if (m.type == MTR_BSDF_ROUGHDIELECTRIC) { rough_dielectric_eval_pdf(m, wi, wo, val, pdf); return; }
    if (m.type == MTR_BSDF_PLASTIC) { plastic_eval_pdf(m, albedo, wi, wo, val, pdf); return; }
    val = mk(0, 0, 0); pdf = 0.0f;
    const float ci = wi.z, co = wo.z;
    if (!(ci > 0.0f && co > 0.0f)) return;
    const f3 H = normalize(mk(wo.x + wi.x, wo.y + wi.y, wo.z + wi.z));
    const bool beck = (m.flags & MTR_MAT_BECKMANN) != 0u;
    const float au = m.alpha, av = rough_alpha_v(m);
    const float D = mf_eval(H, au, av, beck);
    const float g1i = mf_smith_g1(wi, H, au, av, beck);
    if (m.type == MTR_BSDF_ROUGHCONDUCTOR) {
        const float wih = dot(wi, H);
        if (wih > 0.0f && dot(wo, H) > 0.0f) pdf = (D * g1i) / (4.0f * ci);
        if (D != 0.0f) {
            const float G = g1i * mf_smith_g1(wo, H, au, av, beck);
            const float r = (D * G) / (4.0f * ci);
            val = mk((r * fresnel_conductor(wih, m.a[0], m.b[0])) * m.c[0], (r * fresnel_conductor(wih, m.a[1], m.b[1])) * m.c[1],
                     (r * fresnel_conductor(wih, m.a[2], m.b[2])) * m.c[2]);
        }
        return;
    }
    // roughplastic
    const float t_i = rough_transmittance(m, ci), t_o = rough_transmittance(m, co);
    float ps = (1.0f - t_i) * m.specular_sampling_weight, pdif = t_i * (1.0f - m.specular_sampling_weight);
    ps = ps / (ps + pdif); pdif = 1.0f - ps;
    pdf = fmaf(pdif, kInvPi * co, ((D * g1i) / (4.0f * ci)) * ps);
    float F, ct, eit, eti;
    fresnel_dielectric(dot(wi, H), m.int_ior / m.ext_ior, F, ct, eit, eti);
    const float G = g1i * mf_smith_g1(wo, H, au, av, beck);
    const float spec = ((F * D) * G) / (4.0f * ci);
    const float eta = m.int_ior / m.ext_ior, inv_eta_2 = 1.0f / (eta * eta);
    const float dscale = (((kInvPi * inv_eta_2) * co) * t_i) * t_o;
    const float a[3] = { albedo.x, albedo.y, albedo.z };
    float o[3];
    for (int k = 0; k < 3; ++k) {
        const float diff = a[k] / (1.0f - ((m.flags & MTR_MAT_NONLINEAR) ? a[k] * m.internal_reflectance : m.internal_reflectance));
        o[k] = fmaf(diff, dscale, spec * m.c[k]);
    }
    val = mk(o[0], o[1], o[2]);
}

struct BsdfSample { f3 wo; float pdf, eta; bool delta; f3 w; };
// [Plastic::sample]: the coat's mirror direction with probability ps (a delta lobe), else a cosine-weighted direction of the base
MTR_HD void plastic_sample(const mtr_material &m, f3 albedo, f3 wi, float u1, float ua, float ub, BsdfSample &bs)
{
    const float ci = wi.z;
    if (!(ci > 0.0f)) return;
    const float eta = m.int_ior / m.ext_ior, inv_eta_2 = 1.0f / (eta * eta);
    float f_i, ct, eit, eti;
    fresnel_dielectric(ci, eta, f_i, ct, eit, eti);
    float ps, pdif;
    plastic_probs(m, f_i, ps, pdif);
    if (u1 < ps) {
        bs.wo = mk(-wi.x, -wi.y, wi.z); bs.pdf = ps; bs.delta = true;
        const float s = f_i / ps;
        bs.w = mk(m.c[0] * s, m.c[1] * s, m.c[2] * s);
    } else {
        const f3 wo = cosine_hemisphere(ua, ub);
        float f_o;
        fresnel_dielectric(wo.z, eta, f_o, ct, eit, eti);
        bs.wo = wo; bs.pdf = pdif * (kInvPi * wo.z);
        const float scale = ((inv_eta_2 * (1.0f - f_i)) * (1.0f - f_o)) / pdif;
        const float a[3] = { albedo.x, albedo.y, albedo.z };
        float o[3];
        for (int k = 0; k < 3; ++k)
            o[k] = (a[k] / (1.0f - ((m.flags & MTR_MAT_NONLINEAR) ? a[k] * m.internal_reflectance : m.internal_reflectance))) * scale;
        if (bs.pdf > 0.0f) bs.w = mk(o[0], o[1], o[2]);
    }
}
// [ThinDielectric::sample] a thin slab: no refraction, the internal reflections folded into r' = 2 r / (1 + r); two delta lobes
MTR_HD void thin_dielectric_sample(const mtr_material &m, f3 wi, float u1, BsdfSample &bs)
{
    float r, ct, eit, eti;
    fresnel_dielectric(fabsf(wi.z), m.int_ior / m.ext_ior, r, ct, eit, eti);
    if (r < 1.0f) r *= 2.0f / (1.0f + r);
    const bool refl = u1 <= r;
    bs.delta = true; bs.eta = 1.0f;
    bs.pdf = refl ? r : 1.0f - r;
    if (refl) { bs.wo = mk(-wi.x, -wi.y, wi.z); bs.w = mk(m.c[0], m.c[1], m.c[2]); }
    else { bs.wo = mk(-wi.x, -wi.y, -wi.z); bs.w = mk(m.c2[0], m.c2[1], m.c2[2]); }
}
// [RoughConductor::sample, RoughPlastic::sample]; wi on the two-sided side
// [RoughDielectric::sample]: a visible normal for wi flipped to the upper side, reflection with probability F, else refraction
MTR_HD void rough_dielectric_sample(const mtr_material &m, f3 wi, float u1, float ua, float ub, BsdfSample &bs)
{
    const float ci = wi.z;
    if (ci == 0.0f) return;
    const bool beck = (m.flags & MTR_MAT_BECKMANN) != 0u;
    const float au = m.alpha, av = rough_alpha_v(m);
    float pdf_m;
    const f3 mm = ggx_sample(mulsign3(wi, ci), au, av, ua, ub, pdf_m, beck);
    if (pdf_m == 0.0f) return;
    const float wim = dot(wi, mm);
    float F, ct, eit, eti;
    fresnel_dielectric(wim, m.int_ior / m.ext_ior, F, ct, eit, eti);
    const bool refl = u1 <= F;
    float pdf = pdf_m * (refl ? F : 1.0f - F);
    f3 wo, w; float dwh_dwo;
    if (refl) {
        wo = mk(fmaf(mm.x, 2.0f * wim, -wi.x), fmaf(mm.y, 2.0f * wim, -wi.y), fmaf(mm.z, 2.0f * wim, -wi.z));     // reflect(wi, m)
        bs.eta = 1.0f;
        w = mk(m.c[0], m.c[1], m.c[2]);
        dwh_dwo = 1.0f / (4.0f * dot(wo, mm));
    } else {
        const float k = fmaf(wim, eti, ct);                                                                        // refract(wi, m, cos_theta_t, eta_ti)
        wo = mk(fmaf(mm.x, k, -(wi.x * eti)), fmaf(mm.y, k, -(wi.y * eti)), fmaf(mm.z, k, -(wi.z * eti)));
        bs.eta = eit;
        const float f2 = eti * eti;                                                                                // radiance: solid-angle compression
        w = mk(m.c2[0] * f2, m.c2[1] * f2, m.c2[2] * f2);
        const float wom = dot(wo, mm), t = fmaf(eit, wom, wim);
        dwh_dwo = ((eit * eit) * wom) / (t * t);
    }
    const float g1 = mf_smith_g1(wo, mm, au, av, beck);
    bs.wo = wo;
    bs.pdf = pdf * fabsf(dwh_dwo);
    bs.w = mk(w.x * g1, w.y * g1, w.z * g1);
}

MTR_HD void rough_sample(const mtr_material &m, f3 albedo, f3 wi, float u1, float ua, float ub, BsdfSample &bs)
{
    if (m.type == MTR_BSDF_ROUGHDIELECTRIC) { rough_dielectric_sample(m, wi, u1, ua, ub, bs); return; }
    if (m.type == MTR_BSDF_PLASTIC) { plastic_sample(m, albedo, wi, u1, ua, ub, bs); return; }
    const float ci = wi.z;
    if (!(ci > 0.0f)) return;
    const bool beck = (m.flags & MTR_MAT_BECKMANN) != 0u;
    if (m.type == MTR_BSDF_ROUGHCONDUCTOR) {
        float pdf;
        const f3 mm = ggx_sample(wi, m.alpha, rough_alpha_v(m), ua, ub, pdf, beck);
        const float wim = dot(wi, mm);
        const f3 wo = mk(fmaf(mm.x, 2.0f * wim, -wi.x), fmaf(mm.y, 2.0f * wim, -wi.y), fmaf(mm.z, 2.0f * wim, -wi.z));     // reflect(wi, m)
        bs.wo = wo;
        const bool ok = (pdf != 0.0f) && (wo.z > 0.0f);
        const float weight = mf_smith_g1(wo, mm, m.alpha, rough_alpha_v(m), beck);
        bs.pdf = pdf / (4.0f * dot(wo, mm));
        if (ok) bs.w = mk((fresnel_conductor(wim, m.a[0], m.b[0]) * weight) * m.c[0], (fresnel_conductor(wim, m.a[1], m.b[1]) * weight) * m.c[1],
                          (fresnel_conductor(wim, m.a[2], m.b[2]) * weight) * m.c[2]);
        return;
    }
    const float t_i = rough_transmittance(m, ci);
    float ps = (1.0f - t_i) * m.specular_sampling_weight, pdif = t_i * (1.0f - m.specular_sampling_weight);
    ps = ps / (ps + pdif);
    f3 wo;
    if (u1 < ps) {
        float pdf_m;
        const f3 mm = ggx_sample(wi, m.alpha, m.alpha, ua, ub, pdf_m, beck);
        const float wim = dot(wi, mm);
        wo = mk(fmaf(mm.x, 2.0f * wim, -wi.x), fmaf(mm.y, 2.0f * wim, -wi.y), fmaf(mm.z, 2.0f * wim, -wi.z));
    } else wo = cosine_hemisphere(ua, ub);
    bs.wo = wo;
    f3 val; float pdf;
    rough_eval_pdf(m, albedo, wi, wo, val, pdf);
    bs.pdf = pdf;
    if (pdf > 0.0f) { const float ip = 1.0f / pdf; bs.w = mk(val.x * ip, val.y * ip, val.z * ip); }
}

// SCENE TRAITS (round 5): what the host knows about the scene's material / emitter tables, as compile-time bits of the shading
// code — the specialisation Dr.Jit's tracing gives the reference for free (its megakernel only holds the vcall targets the
// scene has).  A trait only REMOVES code whose result is known: every value that is still computed is computed by the same
// operations in the same order, and the folds are exact (x * 1.0f == x), so a specialised kernel returns the bits of the
// general one (tests/test_gpu_parity.py runs both on the same scenes).
//   kTrDiffuse: every material is MTR_BSDF_DIFFUSE without MTR_MAT_TWOSIDED (transientpath.py:157 `si.bsdf(ray)` has ONE
//     target): no delta lobe, so prev_bsdf_delta (:240) is true exactly at depth 0; bs.eta == 1, so eta (:232) stays 1 and
//     `distance += t * eta` (:154), `rr_prob = min(beta_max * eta^2, .95)` (:248) lose their factors; no Fresnel code at all.
//   kTrOneRectEmitter: exactly one emitter, an analytic rectangle (:192 `sample_emitter_direction` has one target): no
//     emitter pick and its sample reuse, no mesh tables, no 1 / n_emitters factors.
//   kTrLeafPair: no triangle leaf of the 8-wide tree holds more than two triangles (the builder's target size; object-space
//     leaves never do): the leaf test is one packed pass without a loop around it.
//   kTrFlatTop: the 8-wide tree is a root whose children are analytic rectangles and at most kFlatMaxBoxes box nodes (FlatTop):
//     the kernels do not walk it (flat_walk_device).
//   kTrFlatLeaves: ... and triangle leaves among them.
//   kTrNoLobes: the scene needs the EXTENDED shading code for interpolated normals or bitmaps only — no material is a GGX / Beckmann lobe,
//     a plastic or a thin dielectric: an extended kernel specialised on it carries none of their code (config 4: the hidden Z's vertex normals;
//     its share 6.82 -> 6.66 ms).
//   kTrGrey: every colour of the scene has three equal channels — reflectances, conductor constants, radiances, the NLOS laser's irradiance —
//     and no bitmap: the three channels of every path run the same instructions on the same operands, so every contribution has r == g == b
//     to the bit.  k_fused<NLOS> then keeps ONE plane of a pixel's row in LDS instead of three (the flush writes it to all three channels):
//     an rgb row of 4096 bins is 48 KB — one row slot per workgroup, no overlap between pixels — a grey one 16 KB: three slots (config 4).
constexpr uint32_t kTrDiffuse = 1u, kTrOneRectEmitter = 2u, kTrLeafPair = 4u, kTrFlatTop = 8u, kTrFlatLeaves = 16u, kTrNoLobes = 32u, kTrGrey = 64u;
template <bool ROUGH, uint32_t TR> constexpr bool lobes_on() { return ROUGH && (TR & kTrNoLobes) == 0u; }
constexpr int flat_kind(uint32_t tr) { return (tr & kTrFlatTop) ? ((tr & kTrFlatLeaves) ? 2 : 1) : 0; }
constexpr uint32_t kTrCornell = kTrDiffuse | kTrOneRectEmitter | kTrLeafPair;      // what the kernels are instantiated for besides 0
constexpr uint32_t kTrCornellFlat = kTrCornell | kTrFlatTop;                       // ... and with the flat top level
constexpr uint32_t kTrFlatFlags = kTrFlatTop | kTrLeafPair;                        // ... the flat top level alone (its box faces are pairs), any materials and emitters
constexpr uint32_t kTrFlatGeneral = kTrFlatFlags | kTrFlatLeaves;                  // ... the instantiation for it: takes top levels with and without triangle leaves

template <bool ROUGH = true, uint32_t TR = 0u>
MTR_HD BsdfSample bsdf_sample(const mtr_material &m, f3 wi, float u1, float ua, float ub, f3 albedo)
{
    BsdfSample bs;
    bs.wo = mk(0, 0, 0); bs.pdf = 0.0f; bs.eta = 1.0f; bs.delta = false; bs.w = mk(0, 0, 0);
    if (TR & kTrDiffuse) {
        bs.wo = cosine_hemisphere(ua, ub);
        bs.pdf = kInvPi * bs.wo.z;
        if (wi.z > 0.0f && bs.pdf > 0.0f) bs.w = mk(m.a[0], m.a[1], m.a[2]);
        return bs;
    }
    bool flip = (m.flags & MTR_MAT_TWOSIDED) && wi.z < 0.0f;
    if (flip) wi.z = -wi.z;
    float ci = wi.z;
    if (m.type == MTR_BSDF_DIFFUSE) {
        bs.wo = cosine_hemisphere(ua, ub);
        bs.pdf = kInvPi * bs.wo.z;
        if (ci > 0.0f && bs.pdf > 0.0f) bs.w = ROUGH ? albedo : mk(m.a[0], m.a[1], m.a[2]);
    } else if (m.type == MTR_BSDF_CONDUCTOR) {
        bs.wo = mk(-wi.x, -wi.y, wi.z); bs.pdf = 1.0f; bs.delta = true;
        if (ci > 0.0f)
            bs.w = mk(m.c[0] * fresnel_conductor(ci, m.a[0], m.b[0]),
                      m.c[1] * fresnel_conductor(ci, m.a[1], m.b[1]),
                      m.c[2] * fresnel_conductor(ci, m.a[2], m.b[2]));
    } else if (m.type == MTR_BSDF_DIELECTRIC) {
        float r, ct, eit, eti;
        fresnel_dielectric(ci, m.int_ior / m.ext_ior, r, ct, eit, eti);
        bool refl = u1 <= r;
        bs.delta = true;
        bs.pdf = refl ? r : 1.0f - r;
        if (refl) { bs.wo = mk(-wi.x, -wi.y, wi.z); bs.w = mk(m.c[0], m.c[1], m.c[2]); }
        else {
            bs.wo = mk(-eti * wi.x, -eti * wi.y, ct); bs.eta = eit;
            float f2 = eti * eti;
            bs.w = mk(m.c2[0] * f2, m.c2[1] * f2, m.c2[2] * f2);
        }
    } else if (lobes_on<ROUGH, TR>() && bsdf_is_rough(m.type)) rough_sample(m, albedo, wi, u1, ua, ub, bs);
    else if (lobes_on<ROUGH, TR>() && m.type == MTR_BSDF_THINDIELECTRIC) thin_dielectric_sample(m, wi, u1, bs);
    if (flip) bs.wo.z = -bs.wo.z;
    return bs;
}

MTR_HD float mis_weight(float a, float b)
{
    float a2 = a * a, b2 = b * b;
    float w = a2 / (a2 + b2);
    return (fabsf(w) <= 3.402823466e+38f) ? w : 0.0f;     // isfinite
}

// ---------------------------------------------------------------- path state
struct Path {
    Ray ray;
    f3 beta, L, prev_p;
    float eta, dist, prev_pdf;
    uint32_t depth;
    uint32_t prev_delta;     // bool
    Rng rng;
    uint32_t px, py;         // film coordinates incl. crop offset (ray generation)
    uint32_t lane;
};

struct BounceStats { uint32_t closest, shadow; };

// lane -> pixel, RNG seeding, jitter, camera ray, loop-state init (transientpath.py:118-131);
// the camera_unwarp pre-pass is done by the caller (it needs a traversal).
MTR_HD void path_begin(Path &p, const Camera &cam, const Film &f, const RenderConst &rc, uint32_t pixel, uint32_t s)
{
    uint32_t lane = pixel * rc.spp_total + s;
    uint32_t py = fastdiv(pixel, rc.div_crop_w), px = pixel - f.crop_w * py;
    p.px = px + f.crop_x; p.py = py + f.crop_y; p.lane = lane;
    p.rng = rng_seed(rc.seed, lane, rc.flags);
    float j1 = rng_f32(p.rng), j2 = rng_f32(p.rng);
    p.ray = camera_ray(cam, rc, p.px, p.py, j1, j2);
    p.beta = mk(1, 1, 1); p.L = mk(0, 0, 0); p.prev_p = mk(0, 0, 0);
    p.eta = 1.0f; p.dist = 0.0f; p.prev_pdf = 1.0f; p.depth = 0; p.prev_delta = 1;
}

// Surface interaction rebuilt from (ray direction, primitive, barycentrics): cheap enough to
// recompute on both sides of the shadow ray instead of keeping 15 registers alive across it.
// sn / ss / stt: the SHADING frame (si.sh_frame); gn: the geometric normal (si.n), which keeps the ray offsets and the
// hidden-geometry cosine of the NLOS tier.  They differ only on smooth-shaded triangles.
struct HitCtx { f3 sp, sn, ss, stt, wi, gn; uint32_t mat, em_plus1; };

// [mitsuba3: SurfaceInteraction::initialize_sh_frame] s = normalize(dp_du - n * dot(n, dp_du)), t = n x s
MTR_HD void sh_frame_of(f3 n, f3 dp_du, f3 &s, f3 &t)
{
    const float dn = dot(n, dp_du);
    s = normalize(mk(fmaf(-n.x, dn, dp_du.x), fmaf(-n.y, dn, dp_du.y), fmaf(-n.z, dn, dp_du.z)));
    t = cross(n, s);
}

// si.p of a hit from its shading record and barycentrics (h[2..4] of TriShade)
MTR_HD f3 hit_point(q4 hc, q4 hd, q4 he, float b1, float b2)
{
    const float b0 = 1.0f - b1 - b2;
    if (fbits(he.w) & kShadeQuadBit)          // rectangle: to_world.transform_affine((u, v, 0)) = fmadd(dv, v, fmadd(du, u, c))
        return mk(fmaf(hd.x, b2, fmaf(hc.y, b1, hd.w)), fmaf(hd.y, b2, fmaf(hc.z, b1, he.x)), fmaf(hd.z, b2, fmaf(hc.w, b1, he.y)));
    return mk(fmaf(hd.w, b0, fmaf(hc.y, b1, hd.x * b2)),
              fmaf(he.x, b0, fmaf(hc.z, b1, hd.y * b2)),
              fmaf(he.y, b0, fmaf(hc.w, b1, hd.z * b2)));
}
// ... of a hit record alone: the vertex a path came from, rebuilt where it is needed (k_wf_shade: an emitter was hit) instead of carried
MTR_HD f3 hit_point(const SceneView &sc, const Hit &h)
{
    const TriShade &tsd = sc.tshade[h.prim];
    return hit_point(tsd.h[2], tsd.h[3], tsd.h[4], h.u, h.v);
}

template <bool SMOOTH = true>
MTR_HD HitCtx hit_ctx(const SceneView &sc, f3 ray_d, const Hit &h)
{
    HitCtx c;
    const TriShade &tsd = sc.tshade[h.prim];
    const q4 ha = tsd.h[0], hb = tsd.h[1], hc = tsd.h[2], hd = tsd.h[3], he = tsd.h[4];
    const float b1 = h.u, b2 = h.v, b0 = 1.0f - b1 - b2;
    c.sp = hit_point(hc, hd, he, b1, b2);
    c.sn = mk(ha.x, ha.y, ha.z); c.ss = mk(ha.w, hb.x, hb.y); c.stt = mk(hb.z, hb.w, hc.x);
    c.gn = c.sn;
    if (SMOOTH && (fbits(he.w) & kShadeSmoothBit)) {
        // [mitsuba3: Mesh::compute_surface_interaction] n = fmadd(n2, b2, fmadd(n1, b1, n0 * b0)), normalised
        const q4 *vn = sc.vnormals + 3u * (uint32_t)h.prim;
        const q4 n0 = vn[0], n1 = vn[1], n2 = vn[2];
        const f3 ns = normalize(mk(fmaf(n2.x, b2, fmaf(n1.x, b1, n0.x * b0)), fmaf(n2.y, b2, fmaf(n1.y, b1, n0.y * b0)),
                                   fmaf(n2.z, b2, fmaf(n1.z, b1, n0.z * b0))));
        c.sn = ns;
        sh_frame_of(ns, mk(ha.w, hb.x, hb.y), c.ss, c.stt);
    }
    const f3 md = -ray_d;
    c.wi = mk(dot(md, c.ss), dot(md, c.stt), dot(md, c.sn));
    const uint32_t mat_em = fbits(he.z);
    c.mat = mat_em & 0xffffu; c.em_plus1 = mat_em >> 16;
    return c;
}

// [mitsuba3: BitmapTexture::eval, filter_type = bilinear, wrap_mode = repeat] uv -> texel space (u w - 1/2, v h - 1/2),
// the four neighbours wrapped by the positive modulo, fmadd(w0.y, fmadd(w0.x, v00, w1.x v10), w1.y fmadd(w0.x, v01, w1.x v11))
MTR_HD f3 texture_eval(const q4 *texels, q4 info, float u, float v)
{
    const int32_t W = (int32_t)fbits(info.y), H = (int32_t)fbits(info.z);
    const float fu = fmaf(u, (float)W, -0.5f), fv = fmaf(v, (float)H, -0.5f);
    const float flu = floorf(fu), flv = floorf(fv);
    const float w1x = fu - flu, w1y = fv - flv, w0x = 1.0f - w1x, w0y = 1.0f - w1y;
    const int32_t ix = (int32_t)flu, iy = (int32_t)flv;
    int32_t x0 = ix % W, x1 = (ix + 1) % W, y0 = iy % H, y1 = (iy + 1) % H;
    x0 += x0 < 0 ? W : 0; x1 += x1 < 0 ? W : 0; y0 += y0 < 0 ? H : 0; y1 += y1 < 0 ? H : 0;
    const q4 *t = texels + fbits(info.x);
    const q4 v00 = t[(size_t)y0 * W + x0], v10 = t[(size_t)y0 * W + x1], v01 = t[(size_t)y1 * W + x0], v11 = t[(size_t)y1 * W + x1];
    const float r0 = fmaf(w0x, v00.x, w1x * v10.x), r1 = fmaf(w0x, v01.x, w1x * v11.x);
    const float g0 = fmaf(w0x, v00.y, w1x * v10.y), g1 = fmaf(w0x, v01.y, w1x * v11.y);
    const float b0 = fmaf(w0x, v00.z, w1x * v10.z), b1 = fmaf(w0x, v01.z, w1x * v11.z);
    return mk(fmaf(w0y, r0, w1y * r1), fmaf(w0y, g0, w1y * g1), fmaf(w0y, b0, w1y * b1));
}
// the colour `a` of a material at a hit: the constant, or its bitmap at the interpolated texture coordinate
// [Mesh::compute_surface_interaction: si.uv = fmadd(uv2, b2, fmadd(uv1, b1, uv0 * b0)); Rectangle: (prim_uv + 1) / 2]
template <bool EXT>
MTR_HD f3 material_albedo(const SceneView &sc, const mtr_material &m, const Hit &h)
{
    if (!EXT || m.albedo_texture == 0u || !sc.texels) return mk(m.a[0], m.a[1], m.a[2]);
    float u, v;
    if (fbits(sc.tshade[h.prim].h[4].w) & kShadeQuadBit) { u = fmaf(h.u, 0.5f, 0.5f); v = fmaf(h.v, 0.5f, 0.5f); }
    else {
        const q4 a = sc.uvs[2u * (uint32_t)h.prim], b = sc.uvs[2u * (uint32_t)h.prim + 1u];
        const float b1 = h.u, b2 = h.v, b0 = 1.0f - b1 - b2;
        u = fmaf(b.x, b2, fmaf(a.z, b1, a.x * b0)); v = fmaf(b.y, b2, fmaf(a.w, b1, a.y * b0));
    }
    return texture_eval(sc.texels, sc.tex_info[m.albedo_texture - 1u], u, v);
}

// [mitsuba3: Interaction::offset_p]
MTR_HD f3 offset_point(f3 sp, f3 sn, f3 dir)
{
    float m = max3(fabsf(sp.x), fabsf(sp.y), fabsf(sp.z));
    float mag = (1.0f + m) * kRayEps;
    if (sign_neg(dot(sn, dir))) mag = -mag;
    return fma3(sn, mag, sp);
}

// What shade_hit leaves pending while the shadow ray is in flight
struct Pending {
    f3 Le;               // emission of this bounce (for L)
    f3 Lr;               // emitter-sampling contribution IF the shadow ray is unoccluded
    float opl;           // its optical path length (distance + ds.dist * eta)
    uint32_t active_next;  // bool: (depth+1 < max_depth) & si.valid
    uint32_t has_shadow;   // bool: a shadow ray must be traced
    // (extended shading) the colour of the material at the hit, when shade_hit has evaluated it: shade_finish samples the BSDF with it
    // instead of walking uv record -> texture record -> four texels a second time (round 6)
    f3 alb; uint32_t has_alb;
};

// Part A of one loop iteration (transientpath.py:148-218): consumes the closest hit, splats the
// emission term, samples the emitter and emits the shadow ray.  RNG: next_2d (:193).
template <bool ROUGH = true, uint32_t TR = 0u, class Sink>
// keep: (optional) the surface interaction for shade_finish — a caller that runs nothing between the two parts (k_wf_shade over scenes
// in HBM: the shadow ray goes to a list) hands it over instead of having shade_finish fetch the shading record a second time
MTR_HD void shade_hit(Path &p, const Hit &h, const SceneView &sc, const Film &film, const RenderConst &rc,
                      Sink &sink, Pending &pd, Ray &shadow, HitCtx *keep = nullptr)
{
    constexpr bool kDiff = (TR & kTrDiffuse) != 0u, kOneRect = (TR & kTrOneRectEmitter) != 0u;
    const bool valid = h.prim >= 0;
    const float eta = kDiff ? 1.0f : p.eta;
    const bool prev_delta = kDiff ? (p.depth == 0u) : (p.prev_delta != 0u);
    const uint32_t n_emitters = kOneRect ? 1u : sc.n_emitters;
    p.dist += kDiff ? h.t : h.t * eta;                               // :154 (inf on a miss)
    pd.active_next = (((p.depth + 1u) < rc.max_depth) & valid) ? 1u : 0u;   // :185
    pd.Le = mk(0, 0, 0); pd.Lr = mk(0, 0, 0); pd.opl = 0.0f; pd.has_shadow = 0u;
    pd.alb = mk(0, 0, 0); pd.has_alb = 0u;
    const uint32_t fx = p.px - film.crop_x, fy = p.py - film.crop_y;       // transient_image_block.py:132
    const bool in_film = (fx < film.width) & (fy < film.height);
    float u1 = rng_f32(p.rng), u2 = rng_f32(p.rng);                  // :193, unconditional for a live lane
    if (!valid) return;
    const HitCtx c = hit_ctx<ROUGH>(sc, p.ray.d, h);
    if (keep) *keep = c;
    const mtr_material &mat = sc.mats[c.mat];

    // direct emission (:166-176)
    if (c.em_plus1 != 0u && !(rc.flags & MTR_FLAG_DISCARD_DIRECT_LIGHT)) {
        const Emitter &E = sc.ems[kOneRect ? 0u : c.em_plus1 - 1u];
        f3 rel = c.sp - p.prev_p;
        float dist = sqrtf(dot(rel, rel));
        f3 dd = rel / dist;
        float em_pdf = 0.0f;
        if (!prev_delta) {
            float dp = dot(dd, c.sn);                  // DirectionSample(scene, si, ref): ds.n = si.sh_frame.n
            if (dp < 0.0f) {
                float adp = fabsf(dp);
                em_pdf = E.inv_area * (adp != 0.0f ? (dist * dist) / adp : 0.0f);
                if (n_emitters > 1) em_pdf *= rc.inv_n_emitters;
            }
        }
        float mis = mis_weight(p.prev_pdf, em_pdf);
        if (c.wi.z > 0.0f)
            pd.Le = mk((p.beta.x * mis) * E.radiance[0], (p.beta.y * mis) * E.radiance[1],
                       (p.beta.z * mis) * E.radiance[2]);
        // add_transient(Le, distance) :179-180; common.py:417-421 pre-multiplies the sample scale
        float vr = pd.Le.x * rc.sample_scale, vg = pd.Le.y * rc.sample_scale, vb = pd.Le.z * rc.sample_scale;
        if (in_film && (vr != 0.0f || vg != 0.0f || vb != 0.0f)) {       // adding +0 is a no-op
            int32_t bin = film_bin(film, p.dist);
            if (bin >= 0) sink.splat(fx, fy, (uint32_t)bin, vr, vg, vb, p.dist, p.depth, 0u);
        }
    }

    // emitter sampling (:188-213); only smooth BSDFs (diffuse, the rough lobes) take part
    if (pd.active_next && (kDiff || mat.type == MTR_BSDF_DIFFUSE || (lobes_on<ROUGH, TR>() && bsdf_is_rough(mat.type))) && n_emitters > 0 ) {
        uint32_t ei = 0;
        if (n_emitters > 1) {
            float su = u1 * rc.n_emitters_f;
            uint32_t i = (uint32_t)su;
            if (i > n_emitters - 1) i = n_emitters - 1;
            ei = i; u1 = su - (float)i;
        }
        const Emitter &E = sc.ems[ei];
        f3 ep, en;
        if (!kOneRect && E.is_mesh) {
            mesh_sample_position(sc.samp_tris, sc.face_cdf, sc.face_pmf, E.first_tri, E.n_tris, u1, u2, ep, en, ROUGH ? sc.samp_vn : nullptr);
        } else {
            float a = fmaf(u1, 2.0f, -1.0f), b = fmaf(u2, 2.0f, -1.0f);
            ep = mk(fmaf(E.du[0], a, fmaf(E.dv[0], b, E.center[0])),
                    fmaf(E.du[1], a, fmaf(E.dv[1], b, E.center[1])),
                    fmaf(E.du[2], a, fmaf(E.dv[2], b, E.center[2])));
            en = ld3(E.n);
        }
        f3 dd = ep - c.sp;
        float dist2 = dot(dd, dd), dist = sqrtf(dist2);
        dd = dd / dist;
        float dp = dot(dd, en), adp = fabsf(dp);
        float x = dist2 / adp;
        float pdf_dir = E.inv_area * ((fabsf(x) <= 3.402823466e+38f) ? x : 0.0f);
        if ((dp < 0.0f) & (pdf_dir != 0.0f)) {
            f3 emw = ld3(E.radiance) / pdf_dir;
            float pdf = pdf_dir;
            if (n_emitters > 1) { pdf = pdf_dir * rc.inv_n_emitters; emw = emw * rc.n_emitters_f; }
            if (pdf != 0.0f) {
                // BSDF value * cos and MIS (:207-213), evaluated before the visibility test
                f3 wo = mk(dot(dd, c.ss), dot(dd, c.stt), dot(dd, c.sn));
                f3 wi_e = c.wi;
                if (!kDiff && (mat.flags & MTR_MAT_TWOSIDED) && wi_e.z < 0.0f) { wi_e.z = -wi_e.z; wo.z = -wo.z; }
                // shadow ray: spawn_ray_to(ds.p) + ray_test
                f3 so = offset_point(c.sp, c.gn, ep - c.sp);
                f3 sd = ep - so;
                float sdist = sqrtf(dot(sd, sd));
                shadow.o = so; shadow.d = sd / sdist; shadow.tmax = sdist * (1.0f - kShadowEps);
                pd.has_shadow = 1u;
                if (lobes_on<ROUGH, TR>() && bsdf_is_rough(mat.type)) {
                    f3 bval; float bpdf;
                    pd.alb = material_albedo<ROUGH>(sc, mat, h); pd.has_alb = 1u;
                    rough_eval_pdf(mat, pd.alb, wi_e, wo, bval, bpdf);
                    float mis_em = mis_weight(pdf, bpdf);
                    pd.Lr = mk(((p.beta.x * mis_em) * bval.x) * emw.x, ((p.beta.y * mis_em) * bval.y) * emw.y,
                               ((p.beta.z * mis_em) * bval.z) * emw.z);
                    pd.opl = p.dist + dist * eta;
                } else
                if (wi_e.z > 0.0f && wo.z > 0.0f) {
                    float bpdf = kInvPi * wo.z;
                    float mis_em = mis_weight(pdf, bpdf);
                    const f3 alb = material_albedo<ROUGH>(sc, mat, h);
                    if (ROUGH) { pd.alb = alb; pd.has_alb = 1u; }
                    pd.Lr = mk(((p.beta.x * mis_em) * ((alb.x * kInvPi) * wo.z)) * emw.x,
                               ((p.beta.y * mis_em) * ((alb.y * kInvPi) * wo.z)) * emw.y,
                               ((p.beta.z * mis_em) * ((alb.z * kInvPi) * wo.z)) * emw.z);
                    pd.opl = p.dist + (kDiff ? dist : dist * eta);   // :217
                }
            }
        }
    }
}

// Part B (transientpath.py:216-257, :318): commits the emitter-sampling term given the shadow-ray
// answer, samples the BSDF, updates the loop state and applies Russian roulette.
// RNG: next_1d, next_2d (:223-224), next_1d (:256).  Returns active_next.
template <bool ROUGH = true, uint32_t TR = 0u, class Sink>
MTR_HD bool shade_finish(Path &p, const Hit &h, bool occluded, const Pending &pd, const SceneView &sc,
                         const Film &film, const RenderConst &rc, Sink &sink, const HitCtx *kept = nullptr)
{
    constexpr bool kDiff = (TR & kTrDiffuse) != 0u;
    const bool valid = h.prim >= 0;
    bool active_next = pd.active_next != 0u;
    f3 Lr = mk(0, 0, 0);
    if (pd.has_shadow && !occluded) {
        Lr = pd.Lr;
        const uint32_t fx = p.px - film.crop_x, fy = p.py - film.crop_y;
        float vr = Lr.x * rc.sample_scale, vg = Lr.y * rc.sample_scale, vb = Lr.z * rc.sample_scale;
        if ((fx < film.width) & (fy < film.height) && (vr != 0.0f || vg != 0.0f || vb != 0.0f)) {
            int32_t bin = film_bin(film, pd.opl);
            if (bin >= 0) sink.splat(fx, fy, (uint32_t)bin, vr, vg, vb, pd.opl, p.depth, 1u);
        }
    }
    float s1 = rng_f32(p.rng), s2a = rng_f32(p.rng), s2b = rng_f32(p.rng);
    float rr_u = rng_f32(p.rng);

    BsdfSample bs;
    bs.wo = mk(0, 0, 0); bs.pdf = 0.0f; bs.eta = 1.0f; bs.delta = false; bs.w = mk(0, 0, 0);
    f3 sp = mk(0, 0, 0);
    p.L = mk((p.L.x + pd.Le.x) + Lr.x, (p.L.y + pd.Le.y) + Lr.y, (p.L.z + pd.Le.z) + Lr.z);    // :230
    if (valid) {
        const HitCtx c = kept ? *kept : hit_ctx<ROUGH>(sc, p.ray.d, h);
        sp = c.sp;
        if (active_next) {
            const f3 albedo = (ROUGH && pd.has_alb) ? pd.alb : material_albedo<ROUGH>(sc, sc.mats[c.mat], h);
            bs = bsdf_sample<ROUGH, TR>(sc.mats[c.mat], c.wi, s1, s2a, s2b, albedo);     // :222-227
            f3 wo_w = mk(fmaf(c.sn.x, bs.wo.z, fmaf(c.stt.x, bs.wo.y, c.ss.x * bs.wo.x)),
                         fmaf(c.sn.y, bs.wo.z, fmaf(c.stt.y, bs.wo.y, c.ss.y * bs.wo.x)),
                         fmaf(c.sn.z, bs.wo.z, fmaf(c.stt.z, bs.wo.y, c.ss.z * bs.wo.x)));
            p.ray.o = offset_point(c.sp, c.gn, wo_w);                                            // si.spawn_ray :231
            p.ray.d = wo_w;
            p.ray.tmax = kInf;
        }
    }
    if (!kDiff) p.eta *= bs.eta;                                                                 // :232
    p.beta = mk(p.beta.x * bs.w.x, p.beta.y * bs.w.y, p.beta.z * bs.w.z);                        // :233
    p.prev_p = sp; p.prev_pdf = bs.pdf;                                                          // :237-240
    if (!kDiff) p.prev_delta = bs.delta ? 1u : 0u;

    // stopping criterion (:245-257)
    float bmax = max3(p.beta.x, p.beta.y, p.beta.z);
    active_next &= (bmax != 0.0f);
    float rr_prob = fminf(kDiff ? bmax : bmax * (p.eta * p.eta), 0.95f);
    active_next &= rr_prob > 0.0f;
    bool rr_active = p.depth >= rc.rr_depth;
    if (rr_active) {
        float inv = rr_prob > 0.0f ? 1.0f / rr_prob : 0.0f;
        p.beta = p.beta * inv;
    }
    active_next &= (!rr_active) | (rr_u < rr_prob);
    if (valid) p.depth += 1;                                                                     // :318
    return active_next;
}

// One whole iteration of the loop of transientpath.py:140-319, run to completion
// (closest hit -> shade_hit -> shadow ray -> shade_finish).  Returns active_next.
// `refresh(p, sink)` runs after either traversal: a caller that can recompute parts of the path state (k_fused: film
// coordinates, lane id and row slot follow from the sample index) does so there instead of holding them in registers across
// the traversals.
struct NoRefresh { template <class Sink> MTR_HD void operator()(Path &, Sink &) const {} };
// unwarp_here: camera_unwarp (transientpath.py:133-138) without its own traversal — the hit whose distance the unwarp subtracts IS
// bounce 0's closest hit, so distance = -t is set here at depth 0 (bit-identical: -t + t * eta with eta = 1), as k_wf_shade does;
// callers that pass false trace the camera ray themselves before the first bounce
template <bool ROUGH = true, uint32_t TR = 0u, class Stack, class Sink, class Refresh = NoRefresh>
MTR_HD bool path_bounce(Path &p, const SceneView &sc, const Film &film, const RenderConst &rc,
                        Stack &st, Sink &sink, BounceStats &stats, const Refresh &refresh = Refresh(), bool unwarp_here = false)
{
    st.prof_mark(2);
    Hit h = traverse<false, (TR & kTrLeafPair) != 0u, flat_kind(TR)>(sc, p.ray.o, p.ray.d, p.ray.tmax, st);       // :148-151
    st.prof_mark(0);
    stats.closest++;
    Pending pd; Ray shadow;
    shadow.o = mk(0, 0, 0); shadow.d = mk(0, 0, 1); shadow.tmax = 0.0f;
    // (stacks that park path state in LDS: the values come back where they are used, so that they hold no register across
    // the traversals)
    if (Stack::kPark) { p.prev_p = st.unpark_prev_p(); p.prev_pdf = st.unpark_prev_pdf(); p.rng.inc = st.unpark_inc(); }
    refresh(p, sink);
    if (unwarp_here && p.depth == 0u && h.prim >= 0) p.dist = -h.t;
    // (the surface interaction handed over to shade_finish instead of rebuilt — 11 more spilled registers across the shadow walk —
    // was measured on the flat walk in round 6: 53.4 against 53.4 ms)
    shade_hit<ROUGH, TR>(p, h, sc, film, rc, sink, pd, shadow);
    st.prof_mark(1);
    bool occluded = false;
    if (pd.has_shadow) {
        stats.shadow++;
        Hit sh = traverse<true, (TR & kTrLeafPair) != 0u, flat_kind(TR)>(sc, shadow.o, shadow.d, shadow.tmax, st);
        occluded = sh.prim >= 0;
    }
    st.prof_mark(0);
    if (Stack::kPark) p.rng.inc = st.unpark_inc();
    refresh(p, sink);
    const bool an = shade_finish<ROUGH, TR>(p, h, occluded, pd, sc, film, rc, sink);
    if (Stack::kPark) { st.park_prev_p(p.prev_p); st.park_prev_pdf(p.prev_pdf); }
    st.prof_mark(1);
    return an;
}

// ---- the DEFERRED COMMIT of an emitter-sampling term (k_wf_shade on scenes walked in HBM, round 4): the shadow ray is traced after
// the iteration that sampled it and the term is committed first thing in the path's next iteration.  The sums are the reference's
// bit for bit: L = (L + Le) + Lr there; (L + Le) + 0 and + Lr at the commit, before the next vertex's Le.
// (Round 5 tried the same in k_fused, walking a lane's shadow ray and its next closest-hit ray in ONE loop: node wave-steps
// unchanged, primitive wave-steps +72 %, 62.6 -> 87.1 ms — profiles/r05_deferred_walk_experiment.txt.)
// shade_finish's own commit (transientpath.py:216-218, :230), word for word:
template <class Sink>
MTR_HD void commit_pending(f3 &L, f3 Lr, float opl, uint32_t depth_log, uint32_t px, uint32_t py,
                           const Film &film, const RenderConst &rc, Sink &sink)
{
    const uint32_t fx = px - film.crop_x, fy = py - film.crop_y;
    const float vr = Lr.x * rc.sample_scale, vg = Lr.y * rc.sample_scale, vb = Lr.z * rc.sample_scale;
    if ((fx < film.width) & (fy < film.height) && (vr != 0.0f || vg != 0.0f || vb != 0.0f)) {
        const int32_t bin = film_bin(film, opl);
        if (bin >= 0) sink.splat(fx, fy, (uint32_t)bin, vr, vg, vb, opl, depth_log, 1u);
    }
    L = mk(L.x + Lr.x, L.y + Lr.y, L.z + Lr.z);
}

} // namespace mtr
