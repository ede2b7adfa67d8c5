// mtr_wavefront.hip — MTR_MODE_WAVEFRONT: per-bounce kernels over SoA path queues in HBM.
//
// One tile = P pixels x S samples = n_slots lanes; a path keeps its SLOT for life (slot = local
// pixel * S + local sample, so slots are pixel-major and the RNG/lane identity is a pure function
// of the slot).  State lives in 7 SoA planes of float4 (16-byte-per-lane accesses);
// the queues hold slot indices.
//
//   k_wf_raygen   slot -> PCG32 stream, jitter, camera ray, loop-state init       (writes all planes)
//   per bounce:
//   k_wf_trace    live queue -> BVH2 closest hit (node packets staged in LDS)     ray planes -> hit planes
//                 + append of the slot to the queue of its hit MATERIAL TYPE (diffuse / conductor /
//                 dielectric / none / miss): wave64 ballot + popcount + one LDS atomic per (wave, type)
//   k_wf_shade    material-sorted queues -> shade_hit, shadow ray (inline any-hit traversal),
//                 shade_finish; time-bin contributions are appended as 16-byte records to the
//                 per-pixel lists; survivors are compacted into the next live queue
//                 (ballot / prefix popcount / one LDS atomic per wave)
// The queues are SEGMENTED: a segment is a fixed range of `seg` slots covering whole pixels, owned by
// one workgroup per launch; its live list, its five material lists and its pixels' record counters
// are advanced with workgroup-local LDS counters and written back once — there is not a single
// global atomic on the data path (a first version with one global tail counter per queue spent
// its time in ~12 ns same-address atomics: 183 us per launch).
//   k_wf_scatter  the time-bin scatter-add: one workgroup per pixel streams that pixel's records
//                 (coalesced 16 B/lane) into an LDS row histogram and adds the row to the
//                 (H,W,T,4) film once.
#include "mtr_kernels.h"

#include <hip/hip_runtime.h>
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

namespace mtr {

namespace {

template <int DEPTH>
struct WStack {
    static constexpr bool kPark = false;      // (k_fused's stack can park path state in LDS: mtr_kernels.hip)
    __device__ __forceinline__ void park_prev_p(mtr::f3) {}
    __device__ __forceinline__ mtr::f3 unpark_prev_p() const { return mtr::mk(0, 0, 0); }
    __device__ __forceinline__ void park_inc(uint64_t) {}
    __device__ __forceinline__ uint64_t unpark_inc() const { return 0; }
    __device__ __forceinline__ void park_prev_pdf(float) {}
    __device__ __forceinline__ float unpark_prev_pdf() const { return 0.0f; }
    int32_t *base; int sp;
    __device__ __forceinline__ void reset() { sp = 0; }
    __device__ __forceinline__ void push_if(bool c, int32_t v) { base[sp * kBlock] = v; sp += c ? 1 : 0; }
    __device__ __forceinline__ int32_t pop() { --sp; return base[sp * kBlock]; }
    __device__ __forceinline__ bool empty() const { return sp == 0; }
    __device__ __forceinline__ void prof_mark(int) {}
    __device__ __forceinline__ void prof_flat(int) {}
    __device__ __forceinline__ void count(int) {}
    __device__ __forceinline__ void tail(unsigned int) {}
};

__host__ __device__ constexpr uint32_t al16(uint32_t x) { return (x + 15u) & ~15u; }

__device__ __forceinline__ void cp16(void *dst, const void *src, uint32_t bytes, int tid)
{
    const uint4 *s = (const uint4 *)src; uint4 *d = (uint4 *)dst;
    for (uint32_t i = tid; i < bytes / 16u; i += kBlock) d[i] = s[i];
}

// stage the scene in LDS (or point at HBM) and carve the traversal stack
__host__ __device__ inline uint32_t wf_stack_rows(const SceneDev &sc, bool scene_lds) { return (scene_lds ? sc.wide_levels : (sc.wnodes8q ? sc.wide8q_levels : sc.wide4_levels)) + 1u; }

template <int STACK, bool SCENE_LDS>
__device__ __forceinline__ void wf_setup(const SceneDev &sc, unsigned char *smem, int tid, SceneView &sv, WStack<STACK> &st,
                                         uint32_t &off)
{
    // all LDS scratch lives in the dynamic region, every carve offset a multiple of 16 (a static
    // __shared__ in front of it would shift the base and mis-align the ds_read_b128 node fetches)
    off = 64;                                                   // smem[0..63]: workgroup counters
    // the traversal stack holds at most one group per level of the tree that is walked (8-wide in LDS, quantised 4-wide in
    // HBM; +1: push_if writes before it counts) — 15 rows instead of the BVH2's 28 for the staircase; sized exactly,
    // not by the STACK class, so that a depth-27 tree leaves room for 5 workgroups per CU instead of 4
    int32_t *s_stack = (int32_t *)(smem + off); off += wf_stack_rows(sc, SCENE_LDS) * kBlock * 4u;
    sv.n_emitters = sc.n_ems; sv.n_slots = sc.n_slots;
    sv.samp_tris = sc.samp_tris; sv.samp_vn = sc.samp_vn; sv.face_pmf = sc.face_pmf; sv.face_cdf = sc.face_cdf; sv.vnormals = sc.vnormals;
    sv.texels = sc.texels; sv.tex_info = sc.tex_info; sv.uvs = sc.uvs;
    sv.flat_off = (uint32_t)(offsetof(WfArgs, sc) + offsetof(SceneDev, flat));
    if (SCENE_LDS) {
        // a scene staged in LDS is walked through its 8-wide tree (wf_plan), as in k_fused
        WNode *n = (WNode *)(smem + off); off += al16(sc.n_wnodes * sizeof(WNode));
        TriPair *tg = (TriPair *)(smem + off); off += al16(sc.n_slots / 2 * sizeof(TriPair));
        TriShade *ts = (TriShade *)(smem + off); off += al16(sc.n_slots * sizeof(TriShade));
        mtr_material *mm = (mtr_material *)(smem + off); off += al16(sc.n_mats * sizeof(mtr_material));
        Emitter *ee = (Emitter *)(smem + off); off += al16(sc.n_ems * sizeof(Emitter));
        cp16(n, sc.wnodes, al16(sc.n_wnodes * sizeof(WNode)), tid);
        cp16(tg, sc.tpairs, al16(sc.n_slots / 2 * sizeof(TriPair)), tid);
        cp16(ts, sc.tshade, al16(sc.n_slots * sizeof(TriShade)), tid);
        cp16(mm, sc.mats, al16(sc.n_mats * sizeof(mtr_material)), tid);
        cp16(ee, sc.ems, al16(sc.n_ems * sizeof(Emitter)), tid);
        sv.nodes = nullptr; sv.wnodes = n; sv.wnodes4 = nullptr; sv.wnodes8q = nullptr; sv.tpairs = tg; sv.tshade = ts; sv.mats = mm; sv.ems = ee;
        sv.node_pairs = true;
        __syncthreads();
    } else {
        sv.nodes = sc.nodes; sv.tpairs = sc.tpairs; sv.tshade = sc.tshade; sv.mats = sc.mats; sv.ems = sc.ems;
        sv.wnodes = nullptr; sv.wnodes4 = sc.wnodes4; sv.wnodes8q = sc.wnodes8q;
        sv.node_pairs = false;
    }
    st.base = s_stack + tid; st.sp = 0;
}

// Segments are drawn from a ticket counter, not strided over the grid: their live counts differ, and every launch of a
// bounce ends with its slowest workgroup.  Two counters alternate between consecutive launches of the stream: a launch
// draws from a.ticket[a.ticket_cur] and zeroes the other one for its successor.
__device__ __forceinline__ void wf_ticket_begin(const WfArgs &a, int tid)
{
    if (blockIdx.x == 0 && tid == 0) a.ticket[a.ticket_cur ^ 1u] = 0u;
}
// A workgroup's FIRST segment is its own index (no atomic: 2048 workgroups hitting one counter at launch cost a launch of
// a late, nearly empty bounce 90 us), the following ones come from the counter, which therefore counts from gridDim.
// The index runs over the bounce's list of segments that still hold live paths (written by raygen / by the previous bounce's
// shading kernel): with max_depth 65 most launches find a handful of the 32768 segments of a tile alive, and walking all of
// them — a ticket and two barriers each — was 0.37 ms per launch, 520 launches per render.
__device__ __forceinline__ uint32_t wf_next_segment(const WfArgs &a, unsigned char *smem, int tid, bool first)
{
    uint32_t k;
    if (first) k = blockIdx.x;
    else {
        uint32_t *s_sg = (uint32_t *)smem + 15;
        __syncthreads();                  // the previous segment's LDS state is no longer in use
        if (tid == 0) *s_sg = gridDim.x + atomicAdd(a.ticket + a.ticket_cur, 1u);
        __syncthreads();
        k = *s_sg;
    }
    return k < a.seg_list_n[a.parity] ? a.seg_list[(size_t)a.parity * a.n_seg + k] : 0xffffffffu;
}
// the segment keeps `n` live paths for the next bounce: its length, and its place in the next bounce's list
// (n_zombie: paths that have ended but whose last emitter-sampling term still waits for its shadow ray — see k_wf_shade)
__device__ __forceinline__ void wf_segment_survivors(const WfArgs &a, uint32_t sg, uint32_t n, uint32_t n_zombie = 0u)
{
    const uint32_t nxt = a.parity ^ 1u;
    a.seg_live[(size_t)nxt * a.n_seg + sg] = n;
    a.seg_zombie[(size_t)nxt * a.n_seg + sg] = n_zombie;
    if (n + n_zombie) a.seg_list[(size_t)nxt * a.n_seg + atomicAdd(a.seg_list_n + nxt, 1u)] = sg;
}

// ---- SoA-of-quads state: 7 planes of float4 (16 B per lane per access, the coalescing sweet spot;
// also what keeps the gathers through the slot queues efficient) ----------------------------------
//   Q_RAY0 (o.xyz, tmax)   Q_RAY1 (d.xyz, eta)      Q_BETA (beta.xyz, dist)   Q_RAD (L.xyz, prev_pdf)
//   Q_PREV (prev_p.xyz, depth | pending << 30 | prev_delta << 31)   Q_RNG (state lo, hi: 8 B)   Q_HIT (t, u, v, prim)
//   Q_PEND (Lr.xyz, opl): scenes in HBM — the emitter-sampling term of the LAST bounce, waiting for its shadow ray (deferred commit, below)
// Q_RNG is the LAST plane and holds 8 bytes per slot: the PCG32 state.  The stream's increment is a function of (seed, lane)
// (rng_seed) and is recomputed where the state is loaded — 8 bytes less to read and 8 less to write per vertex and bounce
// in kernels that wait on HBM (round 4; the four TEA rounds are hidden by the loads)
// The PATH-TRACER tier keeps less (round 5): Q_BETA and Q_AUX = (PCG32 state lo, hi, prev_pdf, depth | pending << 30 | prev_delta << 31),
// which lives in the plane the NLOS tier calls Q_RAD.  The direction (and eta) of a vertex's incoming ray are read from the ray list
// the trace kernel has just walked (the material lists carry the list position next to the slot); the path's radiance is not carried
// at all: every vertex deposits its own increment into its pixel's steady sum (wave_deposit); and the previous vertex's position
// (read only where a path hits an emitter: the MIS weight of transientpath.py:166-176) is rebuilt from the previous bounce's HIT
// RECORD, which survives because the hit plane alternates with the bounce's parity (Q_HIT / Q_HIT1, the latter in the plane the NLOS
// tier calls Q_PREV).  152 instead of 216 B per vertex in k_wf_shade.
enum Plane { Q_RAY0 = 0, Q_RAY1, Q_BETA, Q_RAD, Q_PREV, Q_HIT, Q_PEND, PL16_COUNT, Q_RNG = PL16_COUNT, Q_AUX = Q_RAD, Q_HIT1 = Q_PREV };
__device__ __forceinline__ int hit_plane(uint32_t parity) { return parity ? (int)Q_HIT1 : (int)Q_HIT; }

// NT: scenes walked in HBM.  Their path state, rays and hits stream through every kernel of a bounce exactly once —
// NON-TEMPORAL accesses keep them from pushing the scene (BVH nodes, triangles) out of L2: staircase, 720 x 1280 x 64 spp,
// 244 ms per render on every run; with ordinary accesses 249 ms on some boxes / runs and 285 - 290 ms on others (same
// binary).  (The same hint on the ray and shadow-ray lists changes nothing.)  With the scene in LDS the state is what L2
// should hold between the kernels of a bounce: ordinary accesses (config 2, wavefront organisation: 155 ms per render,
// 180 - 185 ms with the hint).
template <bool NT>
struct PlanesT {
    float4 *base; uint32_t n;
    __device__ __forceinline__ float4 ld(int pl, uint32_t slot) const { return NT ? nt_load(base + (size_t)pl * n + slot) : base[(size_t)pl * n + slot]; }
    __device__ __forceinline__ void st(int pl, uint32_t slot, float4 v) const
    {
        if (NT) nt_store(base + (size_t)pl * n + slot, v); else base[(size_t)pl * n + slot] = v;
    }
    // the fourth word of a 16-byte plane alone
    __device__ __forceinline__ uint32_t ld_w(int pl, uint32_t slot) const
    {
        const uint32_t *w = (const uint32_t *)(base + (size_t)pl * n + slot) + 3;
        return NT ? __builtin_nontemporal_load(w) : *w;
    }
    __device__ __forceinline__ void st_w(int pl, uint32_t slot, uint32_t v) const
    {
        uint32_t *w = (uint32_t *)(base + (size_t)pl * n + slot) + 3;
        if (NT) __builtin_nontemporal_store(v, w); else *w = v;
    }
    // the 8-byte plane behind the 16-byte ones
    __device__ __forceinline__ uint64_t ld_rng(uint32_t slot) const
    {
        const uint32_t *w = (const uint32_t *)(base + (size_t)PL16_COUNT * n) + 2 * (size_t)slot;
        const uint32_t lo = NT ? __builtin_nontemporal_load(w) : w[0], hi = NT ? __builtin_nontemporal_load(w + 1) : w[1];
        return (uint64_t)lo | ((uint64_t)hi << 32);
    }
    __device__ __forceinline__ void st_rng(uint32_t slot, uint64_t v) const
    {
        uint32_t *w = (uint32_t *)(base + (size_t)PL16_COUNT * n) + 2 * (size_t)slot;
        if (NT) { __builtin_nontemporal_store((uint32_t)v, w); __builtin_nontemporal_store((uint32_t)(v >> 32), w + 1); }
        else { w[0] = (uint32_t)v; w[1] = (uint32_t)(v >> 32); }
    }
};

template <class Planes>
__device__ __forceinline__ Ray load_ray(const Planes &P, uint32_t s, float &eta)
{
    const float4 a = P.ld(Q_RAY0, s), b = P.ld(Q_RAY1, s);
    Ray r; r.o = mk(a.x, a.y, a.z); r.tmax = a.w; r.d = mk(b.x, b.y, b.z); eta = b.w;
    return r;
}
template <class Planes>
// with_origin: also the (origin, tmax) plane.  Only a kernel that traces from the planes reads it (k_wf_nlos_bounce); the
// closest-hit kernel takes its rays from the list in list order, shading needs the direction alone
__device__ __forceinline__ void store_state(const Planes &P, uint32_t s, const Path &p, bool with_origin, uint32_t pend = 0u)
{
    if (with_origin) P.st(Q_RAY0, s, make_float4(p.ray.o.x, p.ray.o.y, p.ray.o.z, p.ray.tmax));
    P.st(Q_RAY1, s, make_float4(p.ray.d.x, p.ray.d.y, p.ray.d.z, p.eta));
    P.st(Q_BETA, s, make_float4(p.beta.x, p.beta.y, p.beta.z, p.dist));
    P.st(Q_RAD, s, make_float4(p.L.x, p.L.y, p.L.z, p.prev_pdf));
    P.st(Q_PREV, s, make_float4(p.prev_p.x, p.prev_p.y, p.prev_p.z, __uint_as_float(p.depth | (pend << 30) | (p.prev_delta << 31))));
    P.st_rng(s, p.rng.state);
}
template <class Planes>
// with_origin = false: shading needs the ray's direction alone (the hit point comes from the barycentrics, the next ray is
// written whole) — the (origin, tmax) plane is not read (16 B per vertex that the data flow of shade_finish kept alive)
__device__ __forceinline__ void load_state(const Planes &P, uint32_t s, Path &p, uint32_t *pend = nullptr, bool with_origin = true)
{
    if (with_origin) p.ray = load_ray(P, s, p.eta);
    else { const float4 b = P.ld(Q_RAY1, s); p.ray.o = mk(0, 0, 0); p.ray.tmax = 0.0f; p.ray.d = mk(b.x, b.y, b.z); p.eta = b.w; }
    const float4 b = P.ld(Q_BETA, s), l = P.ld(Q_RAD, s), v = P.ld(Q_PREV, s);
    p.rng.state = P.ld_rng(s);
    p.beta = mk(b.x, b.y, b.z); p.dist = b.w;
    p.L = mk(l.x, l.y, l.z); p.prev_pdf = l.w;
    p.prev_p = mk(v.x, v.y, v.z);
    const uint32_t fl = __float_as_uint(v.w);
    p.depth = fl & 0x3fffffffu; p.prev_delta = fl >> 31;
    if (pend) *pend = (fl >> 30) & 1u;
    p.rng.inc = 0u;                      // the caller knows the lane: p.rng.inc = rng_inc_of(...)
}

// path-tracer tier (k_wf_shade): what a vertex carries from one bounce to the next — see `enum Plane`.  The ray itself goes to the next
// live list's ray list (origin for the trace kernel, direction and eta for the next k_wf_shade); p.L is the caller's business.
template <class Planes>
__device__ __forceinline__ void store_path(const Planes &P, uint32_t s, const Path &p, uint32_t pend)
{
    P.st(Q_BETA, s, make_float4(p.beta.x, p.beta.y, p.beta.z, p.dist));
    P.st(Q_AUX, s, make_float4(__uint_as_float((uint32_t)p.rng.state), __uint_as_float((uint32_t)(p.rng.state >> 32)), p.prev_pdf,
                               __uint_as_float(p.depth | (pend << 30) | (p.prev_delta << 31))));
}
template <class Planes>
__device__ __forceinline__ void load_path(const Planes &P, uint32_t s, float4 dir_eta, Path &p, uint32_t &pend)
{
    const float4 b = P.ld(Q_BETA, s), x = P.ld(Q_AUX, s);
    p.ray.o = mk(0, 0, 0); p.ray.tmax = 0.0f; p.ray.d = mk(dir_eta.x, dir_eta.y, dir_eta.z); p.eta = dir_eta.w;
    p.beta = mk(b.x, b.y, b.z); p.dist = b.w;
    p.rng.state = (uint64_t)__float_as_uint(x.x) | ((uint64_t)__float_as_uint(x.y) << 32); p.prev_pdf = x.z;
    p.L = mk(0, 0, 0);
    p.prev_p = mk(0, 0, 0);              // (rebuilt by the caller where it is read)
    const uint32_t fl = __float_as_uint(x.w);
    p.depth = fl & 0x3fffffffu; p.prev_delta = fl >> 31; pend = (fl >> 30) & 1u;
    p.rng.inc = 0u;                      // the caller knows the lane: p.rng.inc = rng_inc_of(...)
}

// slot -> (pixel, sample) of the tile
__device__ __forceinline__ void slot_to_lane(const WfArgs &a, uint32_t slot, uint32_t &pixel, uint32_t &s, uint32_t &p_local)
{
    p_local = slot / a.S;
    s = a.spp_begin + (slot - p_local * a.S);
    pixel = a.pix0 + p_local;
}

// ray-direction class used to order the next live list (coherent waves in the next trace):
// octant of the direction (3 bits) and its dominant axis (2 bits)
// ---- wave64 aggregated append: one atomic per (wave, key) ---------------------------------
__device__ __forceinline__ uint32_t wave_append(uint32_t *counter, bool want)
{
    const unsigned long long m = __ballot(want);
    if (m == 0ull) return 0u;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n = (uint32_t)__popcll(m);
    const uint32_t rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    const int leader = __ffsll((long long)m) - 1;
    uint32_t base = 0;
    if ((int)lane == leader) base = atomicAdd(counter, n);
    base = __shfl(base, leader);
    return base + rank;
}

// steady sums of the segment's pixels (LDS, [G][4]): the wave adds up what its lanes bring per distinct pixel (lists are in slot order:
// mostly one pixel per wave) and one lane adds the sums — every vertex deposits its radiance increment, a path that ends its weight
// of 1 (block.put(pos, [L.r, L.g, L.b, 1]), common.py:187-200).  Called by WHOLE waves.
__device__ __forceinline__ void wave_deposit(float *s_steady, bool has, uint32_t p_seg, f3 L, float w)
{
    unsigned long long todo = __ballot(has);
    const uint32_t lane_id = threadIdx.x & 63u;
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t px = __shfl(p_seg, leader);
        const bool mine = has & (p_seg == px);
        float x = mine ? L.x : 0.0f, y = mine ? L.y : 0.0f, z = mine ? L.z : 0.0f, c = mine ? w : 0.0f;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { x += __shfl_xor(x, o); y += __shfl_xor(y, o); z += __shfl_xor(z, o); c += __shfl_xor(c, o); }
        if ((int)lane_id == leader) {
            float *sp = s_steady + 4 * px;
            if (x != 0.0f) __hip_atomic_fetch_add(sp, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (y != 0.0f) __hip_atomic_fetch_add(sp + 1, y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (z != 0.0f) __hip_atomic_fetch_add(sp + 2, z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (c != 0.0f) __hip_atomic_fetch_add(sp + 3, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        todo &= ~__ballot(mine);
    }
}

// TRACE ORDER (WfArgs::q_order): sort key of a ray = (cell of its origin on a 32-cell grid over the scene) << 3 | octant of its direction
__device__ __forceinline__ uint32_t trace_sort_key(const WfArgs &a, f3 o, f3 d)
{
    const uint32_t bx = a.sort_bits[0], by = a.sort_bits[1], bz = a.sort_bits[2];
    const float cx = fminf(fmaxf((o.x - a.sort_lo[0]) * a.sort_scale[0], 0.0f), (float)((1u << bx) - 1u));
    const float cy = fminf(fmaxf((o.y - a.sort_lo[1]) * a.sort_scale[1], 0.0f), (float)((1u << by) - 1u));
    const float cz = fminf(fmaxf((o.z - a.sort_lo[2]) * a.sort_scale[2], 0.0f), (float)((1u << bz) - 1u));
    const uint32_t cell = (((uint32_t)cx << by) | (uint32_t)cy) << bz | (uint32_t)cz;
    return (cell << 3) | (d.x < 0.0f ? 1u : 0u) | (d.y < 0.0f ? 2u : 0u) | (d.z < 0.0f ? 4u : 0u);
}
// counting sort of the n keys (one byte each, LDS) of a segment's list: order[j] = list position of the j-th ray in key order.
// Called by the whole workgroup; s_hist: 260 words of LDS.  (Within one key the order is that of the atomics: it only decides which
// lane traces which ray.)
__device__ __forceinline__ void trace_sort(const uint8_t *s_keys, uint32_t n, uint32_t *s_hist, uint16_t *order, int tid)
{
    s_hist[tid] = 0u;
    __syncthreads();
    for (uint32_t i = tid; i < n; i += kBlock) atomicAdd(&s_hist[s_keys[i]], 1u);
    __syncthreads();
    const uint32_t c = s_hist[tid];
    uint32_t incl = c;
    const uint32_t lane = (uint32_t)tid & 63u;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(incl, o); if (lane >= (uint32_t)o) incl += v; }
    if (lane == 63u) s_hist[256 + (tid >> 6)] = incl;
    __syncthreads();
    uint32_t base = 0u;
    for (int w = 0; w < (tid >> 6); ++w) base += s_hist[256 + w];
    __syncthreads();
    s_hist[tid] = base + incl - c;                       // exclusive prefix = the key's cursor
    __syncthreads();
    for (uint32_t i = tid; i < n; i += kBlock) order[atomicAdd(&s_hist[s_keys[i]], 1u)] = (uint16_t)i;
}

// time-bin contribution -> 16-byte record appended to its pixel's list; list full -> f32 atomics to HBM.
// The list tails of the segment's pixels live in LDS for the duration of the launch.
struct RecordSink {
    uint4 *rec; uint32_t *s_rec_count; uint32_t rec_cap;    // s_rec_count: LDS, indexed by pixel within the segment
    float *film; uint32_t film_w, bins;
    uint32_t n_freq; const float *freq; float start_opl;    // phasor film: records keep the optical path length instead of a bin
    uint32_t p_local, p_seg, lane;                          // pixel within the tile / within the segment
    uint32_t n_splats, n_overflow;
    SplatLog log;
    __device__ __forceinline__ void splat(uint32_t fx, uint32_t fy, uint32_t bin, float r, float g, float b,
                                          float opl, uint32_t depth, uint32_t kind)
    {
        // lanes of a wave that splat together mostly share 1..3 pixels: one LDS atomic per distinct pixel
        unsigned long long todo = __ballot(1);
        const uint32_t lane_id = threadIdx.x & 63u;
        uint32_t idx = 0;
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const uint32_t px = __shfl(p_seg, leader);
            const unsigned long long same = __ballot(p_seg == px) & todo;
            if (p_seg == px) {
                const uint32_t n = (uint32_t)__popcll(same);
                const uint32_t rank = (uint32_t)__popcll(same & ((1ull << lane_id) - 1ull));
                uint32_t base = 0;
                if ((int)lane_id == leader) base = atomicAdd(s_rec_count + px, n);
                base = __shfl(base, leader);
                idx = base + rank;
            }
            todo &= ~same;
        }
        ++n_splats;
        if (idx < rec_cap) {
            rec[(size_t)p_local * rec_cap + idx] = make_uint4(n_freq ? __float_as_uint(opl) : bin, __float_as_uint(r),
                                                              __float_as_uint(g), __float_as_uint(b));
        } else if (n_freq) {
            ++n_overflow;
            float *dst = film + ((size_t)fy * film_w + fx) * (2u * n_freq + 1u);
            const float rel = opl - start_opl;                                 // phasor_hdr_film.py:249
            for (uint32_t f = 0; f < n_freq; ++f) {
                float c, sn;
                phasor_term(freq[f], rel, c, sn);
                unsafeAtomicAdd(dst + 2 * f, r * c); unsafeAtomicAdd(dst + 2 * f + 1, r * sn);
            }
        } else {
            ++n_overflow;
            size_t o = (((size_t)fy * film_w + fx) * bins + bin) * 4u;
            unsafeAtomicAdd(film + o, r); unsafeAtomicAdd(film + o + 1, g); unsafeAtomicAdd(film + o + 2, b);
        }
        if (log.rec) {
            unsigned long long i = atomicAdd(log.count, 1ull);
            if (i < log.cap) {
                uint32_t *R = log.rec + 8 * i;
                R[0] = lane; R[1] = depth | (kind << 16); R[2] = fy * film_w + fx; R[3] = bin;
                R[4] = __float_as_uint(r); R[5] = __float_as_uint(g); R[6] = __float_as_uint(b); R[7] = __float_as_uint(opl);
            }
        }
    }
};

struct NullSink {
    __device__ __forceinline__ void splat(uint32_t, uint32_t, uint32_t, float, float, float, float, uint32_t, uint32_t) {}
};

// ---- kernels ---------------------------------------------------------------------------------
template <int STACK, bool SCENE_LDS>
__global__ void __launch_bounds__(kBlock) k_wf_raygen(const WfArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    SceneView sv; WStack<STACK> st; uint32_t off;
    wf_setup<STACK, SCENE_LDS>(a.sc, smem, tid, sv, st, off);
    const PlanesT<!SCENE_LDS> P{ (float4 *)a.planes, a.n_slots };
    uint32_t n_closest = 0;
    // The path tier's bounce 0 is a function of (pixel, sample) alone: its live list is the identity and k_wf_trace<FIRST> /
    // k_wf_shade<FIRST> REBUILD the camera ray and the path state instead of reading back what this kernel would have written
    // (round 5: 72 + 36 B written here and read there per slot, 58 GB of config 2's 2^28 slots).  The NLOS bounce kernel reads
    // the planes and the lists, so that tier still stores them.
    // camera_unwarp (transientpath.py:133-138: distance = -t of the camera ray's own hit): that hit IS the closest hit of
    // bounce 0, which k_wf_trace is about to find for this very ray — k_wf_shade takes it from there (depth 0) instead of
    // tracing every camera ray twice (config 5: k_wf_raygen 31 -> 6 ms per tile); the ray is still counted
    if (!a.nlos_on && (a.rc.flags & MTR_FLAG_CAMERA_UNWARP) && blockIdx.x == 0 && tid == 0) n_closest = a.n_slots;
    for (uint32_t slot = blockIdx.x * kBlock + tid; a.nlos_on && slot < a.n_slots; slot += gridDim.x * kBlock) {
        uint32_t pixel, s, pl;
        slot_to_lane(a, slot, pixel, s, pl);
        Path p;
        nlos_begin(p, a.nlos, a.film, a.rc, pixel, s);
        store_state(P, slot, p, true);
        a.q_live[slot] = slot;                                   // live queue of bounce 0 = identity
        a.q_ray[2 * (size_t)slot] = make_float4(p.ray.o.x, p.ray.o.y, p.ray.o.z, p.ray.tmax);
        a.q_ray[2 * (size_t)slot + 1] = make_float4(p.ray.d.x, p.ray.d.y, p.ray.d.z, p.eta);
    }
    for (uint32_t sg = blockIdx.x * kBlock + tid; sg < a.n_seg; sg += gridDim.x * kBlock) {
        a.seg_live[sg] = min(a.seg, a.n_slots - sg * a.seg);    // every slot of the segment is live
        a.seg_zombie[sg] = 0u; a.seg_zombie[(size_t)a.n_seg + sg] = 0u;
        a.seg_list[sg] = sg;                                    // ... and every segment is on bounce 0's list (its length: the host)
    }
    if (a.counters) {
        if (n_closest) atomicAdd(&a.counters->rays_closest, (unsigned long long)n_closest);
        if (blockIdx.x == 0 && tid == 0) atomicAdd(&a.counters->paths, (unsigned long long)a.n_slots);
    }
}

// closest hit for the live lists of this bounce; every slot is appended to the list of its hit
// material type inside its segment (sorted-by-material hit queues).
//
// Traversal is PERSISTENT per lane with DYNAMIC FETCH and PHASE VOTING.  Measured on the plain one-ray-per-lane
// while-while loop (in-kernel counters, config 2): a wave spends 73 % of its node steps waiting for its slowest ray
// (sum of per-call wave-maxima / wave steps) — mean / max lane work is 1/3 — and another 1.37x on node runs of
// unequal length; refilling lanes alone does not help (tried first: with every lane busy the node phase still waits for
// the longest of 64 node runs).  So: (1) a lane whose ray is finished takes the next ray of the segment's list from an
// LDS cursor as soon as a quarter of the wave is idle; (2) every iteration the wave executes ONE step of the phase
// that holds more lanes — an inner-node step or a leaf — instead of running each phase to completion.
// Rays finish out of order, so the material lists are built afterwards in list order from the hit materials kept in
// LDS (a scrambled append order costs k_wf_shade its gather coalescing).
#ifndef MTR_WF_REFILL_MIN
#define MTR_WF_REFILL_MIN 16
#endif
#ifndef MTR_WF_TRACE_WAVES_ANY
#define MTR_WF_TRACE_WAVES_ANY 6      // the occlusion instantiation (80 registers, none spilled)
#endif
#ifndef MTR_WF_TRACE_WAVES_LDS
#define MTR_WF_TRACE_WAVES_LDS 1      // scenes staged in LDS: no bound (93 registers, 5 waves per SIMD)
#endif
#ifndef MTR_WF_TRACE_WAVES
#define MTR_WF_TRACE_WAVES 6          // scenes in HBM: the walk waits on loads, 6 waves per SIMD (80 registers, 6 spilled) beat 5 and 8 (measured)
#endif
// ANY: the occlusion pass of a bounce's shadow rays (a.trace_any) as its own instantiation — no hit record to keep, nothing to sort
// into material lists: the compiler drops the closest-hit bookkeeping from the walk instead of carrying both behind a flag
// FIRST: the closest hits of bounce 0 — the live list is the identity and a camera ray is a function of (pixel, sample): both are
// computed here (path_begin, as k_wf_shade<FIRST> does for the rest of the state) instead of written by k_wf_raygen and read back,
// 36 B per slot each way (config 2: 9.7 GB written + 9.7 GB read per render)
template <int STACK, bool SCENE_LDS, bool ANY, bool FIRST = false>
__global__ void __launch_bounds__(kBlock, SCENE_LDS ? MTR_WF_TRACE_WAVES_LDS : (ANY ? MTR_WF_TRACE_WAVES_ANY : MTR_WF_TRACE_WAVES)) k_wf_trace(const WfArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *s_cnt = (uint32_t *)smem;                         // [kWfKeys] list tails of the segment
    uint32_t *s_fetch = (uint32_t *)smem + 8;                   // cursor into the segment's live list
    const int tid = threadIdx.x;
    const uint32_t lane_id = (uint32_t)tid & 63u;
    SceneView sv; WStack<STACK> st; uint32_t off;
    wf_setup<STACK, SCENE_LDS>(a.sc, smem, tid, sv, st, off);
    uint8_t *s_key = (uint8_t *)(smem + off);                   // [seg] hit material type per list position
    const PlanesT<!SCENE_LDS> P{ (float4 *)a.planes, a.n_slots };
    const uint32_t par = a.parity;
#ifdef MTR_PROFILE_SIMT
    unsigned long long prof[4] = { 0, 0, 0, 0 };      // node iterations, lanes in them, leaf iterations, lanes in them
#endif
    constexpr bool any_hit = ANY;                   // occlusion of this bounce's shadow rays instead of closest hits
    wf_ticket_begin(a, tid);
    for (uint32_t sg = wf_next_segment(a, smem, tid, true); sg < a.n_seg; sg = wf_next_segment(a, smem, tid, false)) {
        const uint32_t n_live = any_hit ? a.seg_shadow[sg] : a.seg_live[(size_t)par * a.n_seg + sg];
        if (n_live == 0u) {                  // an emptied segment (most of them in the deep bounces of max_depth 65): no barriers, no LDS
            if (!any_hit && tid < (int)kWfKeys) a.seg_mat[(size_t)sg * kWfKeys + tid] = 0u;
            continue;
        }
        if (tid < (int)kWfKeys) s_cnt[tid] = 0u;
        if (tid == 0) *s_fetch = 0u;
        __syncthreads();
        const uint32_t *q = a.q_live + (size_t)par * a.n_slots + (size_t)sg * a.seg;      // (closest hits: the slots of the material lists)

        Trav tr;
        tr.cur = kTravDone; tr.h.t = kInf; tr.h.u = 0.0f; tr.h.v = 0.0f; tr.h.prim = -1; tr.best_orig = 0xffffffffu;
        tr.o = mk(0, 0, 0); tr.d = mk(0, 0, 1); tr.id = mk(0, 0, 0); tr.noid = mk(0, 0, 0); tr.tmax = 0.0f;
        uint32_t pos = 0;
        bool pending = false;                                   // a finished ray whose result is not written yet
        const float4 *qr = any_hit ? a.r_shadow + 2 * (size_t)sg * a.seg
                                   : a.q_ray + 2 * ((size_t)par * a.n_slots + (size_t)sg * a.seg);   // rays in list order
        const uint16_t *ord = any_hit ? (a.q_order_sh ? a.q_order_sh + (size_t)sg * a.seg : nullptr) : (a.q_order ? a.q_order + (size_t)sg * a.seg : nullptr);
        st.reset();
        for (;;) {
            const bool idle = tr.cur == kTravDone;
            const unsigned long long m_idle = __ballot(idle);
            const uint32_t n_idle = (uint32_t)__popcll(m_idle);
            if (n_idle >= MTR_WF_REFILL_MIN) {
                if (idle && pending) {
                    // RESULTS IN LIST ORDER (round 6).  Hit records and occlusion flags used to be stored by SLOT: 16 bytes / one byte
                    // scattered over the segment's 8192 slots, i.e. partial cache lines that left L2 one by one (config 5: 177 GB written
                    // for 61 GB of hit records, 90 GB for 3 GB of flags).  They are produced in ray-list order and k_wf_shade holds a
                    // vertex's list position next to its slot (q_mat), so they now live at the list position; the flags are gathered in
                    // LDS and leave in one coalesced pass per segment (90 -> 48 GB written by the occlusion pass, the rest of it spilled
                    // registers).  The 16-byte records still reach HBM one by one — rays finish out of order, and there is no LDS left
                    // to gather them in at six workgroups per CU — 184 GB for 61 GB of records, as before.
                    if (any_hit) s_key[pos] = tr.h.prim >= 0 ? (uint8_t)1 : (uint8_t)0;
                    else {
                        // (the planes' streaming store: with an ordinary store the records do NOT meet in L2 either — 245 against 184 GB written
                        // per config-5 render and 0.7 % more time, measured; only the flags, gathered in LDS, leave as whole lines)
                        P.st(hit_plane(par), sg * a.seg + pos, make_float4(tr.h.t, tr.h.u, tr.h.v, __uint_as_float((uint32_t)tr.h.prim)));
                        // hit: the list key rides in the low bits of the tie-break word of the best hit (TriPair, mtr_core.h)
                        const uint32_t key = tr.h.prim >= 0 ? (tr.best_orig & 7u) : 4u;
                        s_key[pos] = (uint8_t)key;
                    }
                    pending = false;
                }
                const int leader = __ffsll((long long)m_idle) - 1;
                uint32_t base = 0;
                if ((int)lane_id == leader) base = atomicAdd(s_fetch, n_idle);
                base = __shfl(base, leader);
                const uint32_t idx = base + (uint32_t)__popcll(m_idle & ((1ull << lane_id) - 1ull));
                if (idle && idx < n_live) {
                    // TRACE ORDER: the idx-th ray to trace sits at list position ord[idx] (bounce 0's camera rays are coherent as they are)
                    pos = (!FIRST && ord) ? (uint32_t)ord[idx] : idx;
                    if (FIRST) {
                        uint32_t pixel, s, pl;
                        slot_to_lane(a, sg * a.seg + idx, pixel, s, pl);
                        Path p;
                        path_begin(p, a.cam, a.film, a.rc, pixel, s);
                        trav_init(tr, sv, p.ray.o, p.ray.d, p.ray.tmax, st);
                    } else {
                        const float4 r0 = qr[2 * (size_t)pos], r1 = qr[2 * (size_t)pos + 1];
                        // (a bounce ray's tmax is infinite — shade_finish — and the word of the list that would hold it carries the list
                        // position of the path's PREVIOUS vertex instead: k_wf_shade, `prev_pos`)
                        trav_init(tr, sv, mk(r0.x, r0.y, r0.z), mk(r1.x, r1.y, r1.z), any_hit ? r0.w : kInf, st);
                    }
                    pending = true;
                }
            }
            const bool at_node = tr.cur >= 0, at_leaf = (tr.cur < 0) & (tr.cur != kTravDone);
            const uint32_t n_node = (uint32_t)__popcll(__ballot(at_node)), n_leaf = (uint32_t)__popcll(__ballot(at_leaf));
            if (n_node + n_leaf == 0u) break;                   // nothing running and nothing left to fetch
#ifdef MTR_PROFILE_SIMT
            if (lane_id == 0) { if (n_node >= n_leaf) { prof[0] += 1; prof[1] += n_node; } else { prof[2] += 1; prof[3] += n_leaf; } }
#endif
            if (SCENE_LDS) {
                if (n_node >= n_leaf) { if (at_node) wide_node_step<kWide, true>(tr, sv.wnodes, st); }
                else { if (at_leaf) wide_leaf_step<kWide>(tr, sv, sv.wnodes, st, any_hit); }
            } else {
                // (release builds always hold the quantised 8-wide tree of a scene walked in HBM — mtr_scene_host.cpp — so the
                // 4-wide walker is compiled only where a knob can switch the tree off: half the code, no spills in either instantiation)
#ifdef MTR_EXPERIMENTS
                if (sv.wnodes8q) {
#else
                {
#endif
                    if (n_node >= n_leaf) { if (at_node) q8_node_step(tr, sv.wnodes8q, st); }
                    else { if (at_leaf) q8_leaf_step(tr, sv, st, any_hit); }
                }
#ifdef MTR_EXPERIMENTS
                else {
                    if (n_node >= n_leaf) { if (at_node) qwide_node_step(tr, sv.wnodes4, st); }
                    else { if (at_leaf) qwide_leaf_step(tr, sv, st, any_hit); }
                }
#endif
            }
        }
        __syncthreads();
        if (any_hit) {       // the occlusion flags of the segment's shadow list, in list order: 16 bytes per lane
            uint4 *dst = (uint4 *)(a.occ + (size_t)sg * a.occ_stride);
            for (uint32_t i = tid; 16u * i < n_live; i += kBlock) dst[i] = ((const uint4 *)s_key)[i];
        }
        // material lists in list order
        const uint32_t n_round = any_hit ? 0u : (n_live + 63u) & ~63u;     // whole waves stay in the loop (ballots)
        for (uint32_t i = tid; i < n_round; i += kBlock) {
            const bool on = i < n_live;
            const uint32_t key = on ? (uint32_t)s_key[i] : kWfKeys;
            const uint32_t sl = on ? (FIRST ? sg * a.seg + i : sg * a.seg + (q[i] & 0xffffu)) : 0u;      // (a live-list entry: shadow-list position << 16 | slot within the segment)
#pragma unroll
            for (uint32_t k = 0; k < kWfKeys; ++k) {
                const bool mine = on & (key == k);
                if (__ballot(mine) != 0ull) {
                    const uint32_t p2 = wave_append(&s_cnt[k], mine);
                    if (mine) a.q_mat[(size_t)k * a.n_slots + (size_t)sg * a.seg + p2] = (i << 16) | (sl - sg * a.seg);     // (list position, slot within the segment): seg <= 2^16, wf_render
                }
            }
        }
        __syncthreads();
        if (!any_hit && tid < (int)kWfKeys) a.seg_mat[(size_t)sg * kWfKeys + tid] = s_cnt[tid];
        __syncthreads();
    }
#ifdef MTR_PROFILE_SIMT
    if (lane_id == 0 && a.counters) {
        atomicAdd(&a.counters->r0, (prof[0] << 32) | (prof[1] >> 6));      // node iterations | lanes / 64
        atomicAdd(&a.counters->r1, (prof[2] << 32) | (prof[3] >> 6));
    }
#endif
}

// shade the material-sorted lists of every segment; survivors form the next live list
// waves per SIMD the register allocator leaves room for: a scene in HBM/L2 needs the occupancy to hide its latency
// (4: 425 ms vs 3: 452 ms on the staircase); with the scene in LDS the 168 registers of 3 waves avoid 29 spilled
// dwords (187 ms vs 215 ms per config-2 render)
// EXT: the extended shading code (GGX lobes, interpolated normals) — only scenes that need it pay for it
// (config 2, wavefront organisation: k_wf_shade 1.44 ms per launch without, 1.82 ms with)
//
// DEFERRED COMMIT of the emitter-sampling term (scenes walked in HBM; round 4).  The reference's iteration is
// closest hit -> shade_hit (emission, emitter sample, SHADOW RAY) -> shade_finish (commit the sample if unoccluded, BSDF sample,
// roulette).  Tracing the shadow ray inside this kernel was 83 % of it (lanes waiting for the slowest ray of their wave), so
// rounds 1-3 ran shade_hit TWICE: k_wf_shadow_gen (null sink: only the shadow rays) -> k_wf_trace (occlusion) -> k_wf_shade —
// 160 B per vertex and a second random fetch of the 80-byte shading record just to learn the ray.  Nothing in shade_finish but
// the commit depends on the shadow ray, so the iteration is now run ONCE: shade_hit writes the shadow ray to the segment's
// shadow list and parks the term (Lr, optical path length) in the Q_PEND plane, shade_finish runs with the term withheld,
// the occlusion kernel follows, and the NEXT bounce's k_wf_shade commits the parked term first thing.  Every TRANSIENT record —
// value, optical path length, bin — is bit for bit the reference's.  The STEADY image is not summed in the reference's order
// since round 5: a path no longer carries its radiance L = (L + Le) + Lr from vertex to vertex; every vertex deposits its own
// increment into its pixel's sum (wave_deposit: a wave reduction per distinct pixel, then LDS float atomics), so the steady image
// of this organisation agrees with k_fused's and the oracle's to f32 summation order (1e-6 of its norm at 1024 spp), not to the bit.
// A path that ENDS with a term parked becomes a "zombie" (its depth in Q_AUX.w, the term in Q_PEND, its slot on the
// segment's zombie list); the next launch commits it and counts the path in its pixel's steady sum.  No term can be left at the end of a render:
// a vertex samples the emitter only if depth + 1 < max_depth, and the host's live count includes the zombies.
// (commit_pending: mtr_core.h — shade_finish's own commit, shared with k_fused's deferred organisation)

#ifndef MTR_WF_SHADE_WAVES
#define MTR_WF_SHADE_WAVES 4
#endif
#ifndef MTR_WF_SHADE_WAVES_LDS
#define MTR_WF_SHADE_WAVES_LDS 3
#endif
// TR: scene traits (mtr_core.h kTr*; scenes staged in LDS only): shading code the scene's tables cannot reach is not compiled in
// FIRST: the launch that shades bounce 0 — the path state is rebuilt from (pixel, sample) instead of loaded (k_wf_raygen); its own
// instantiation, so that the camera and path_begin's code stay out of the kernel every other bounce runs
template <int STACK, bool SCENE_LDS, bool EXT, uint32_t TR = 0u, bool FIRST = false>
__global__ void __launch_bounds__(kBlock, SCENE_LDS ? MTR_WF_SHADE_WAVES_LDS : MTR_WF_SHADE_WAVES) k_wf_shade(const WfArgs a)
{
    constexpr bool DEFER = !SCENE_LDS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *s_next_p = (uint32_t *)smem;                      // tail of the segment's next live list
    uint32_t *s_shadow_p = (uint32_t *)smem + 1;                // (DEFER) tail of the segment's shadow-ray list
    uint32_t *s_zombie_p = (uint32_t *)smem + 2;                // (DEFER) tail of the segment's next zombie list
    const int tid = threadIdx.x;
    SceneView sv; WStack<STACK> st; uint32_t off;
    wf_setup<STACK, SCENE_LDS>(a.sc, smem, tid, sv, st, off);
    uint32_t *s_rec = (uint32_t *)(smem + off);                 // [G] record-list tails of the segment's pixels
    float *s_steady = (float *)(smem + off + al16(a.G * 4u));   // [G][4] radiance sums of the paths that end here
    uint8_t *s_sortkey = (uint8_t *)(smem + off + al16(a.G * 4u) + al16(a.G * 16u));      // [seg] TRACE ORDER: keys of the next live list
    uint8_t *s_sortkey_sh = s_sortkey + al16(a.seg);            // [seg] ... of the shadow list (DEFER)
    uint32_t *s_hist = (uint32_t *)(s_sortkey_sh + (DEFER ? al16(a.seg) : 0u));           // [260]
    const PlanesT<!SCENE_LDS> P{ (float4 *)a.planes, a.n_slots };
    const uint32_t par = a.parity;
    uint32_t n_closest = 0, n_shadow = 0, n_bounce = 0, n_splats = 0, n_over = 0, n_alive = 0;
    wf_ticket_begin(a, tid);
    for (uint32_t sg = wf_next_segment(a, smem, tid, true); sg < a.n_seg; sg = wf_next_segment(a, smem, tid, false)) {
        const uint32_t pl0 = sg * a.G;                          // first pixel (tile-local) of the segment
        const uint32_t npx = min(a.G, a.P - pl0);
        const uint32_t n_z = DEFER ? a.seg_zombie[(size_t)par * a.n_seg + sg] : 0u;
        {   // an emptied segment: nothing to shade, nothing survives
            uint32_t n_all = n_z;
            for (uint32_t k = 0; k < kWfKeys; ++k) n_all += a.seg_mat[(size_t)sg * kWfKeys + k];
            if (n_all == 0u) { if (tid == 0) { wf_segment_survivors(a, sg, 0u); if (DEFER) a.seg_shadow[sg] = 0u; } continue; }
        }
        for (uint32_t t = tid; t < npx; t += kBlock) s_rec[t] = a.rec_count[pl0 + t];
        for (uint32_t t = tid; t < 4 * npx; t += kBlock) s_steady[t] = 0.0f;
        if (tid == 0) { *s_next_p = 0u; *s_shadow_p = 0u; *s_zombie_p = 0u; }
        __syncthreads();
        uint32_t *q_next = a.q_live + (size_t)(par ^ 1u) * a.n_slots + (size_t)sg * a.seg;
        float4 *r_next = a.q_ray + 2 * ((size_t)(par ^ 1u) * a.n_slots + (size_t)sg * a.seg);
        const float4 *r_cur = a.q_ray + 2 * ((size_t)par * a.n_slots + (size_t)sg * a.seg);      // the rays k_wf_trace has just walked, in list order
        uint32_t *qz_next = DEFER ? a.q_zombie + (size_t)(par ^ 1u) * a.n_slots + (size_t)sg * a.seg : nullptr;
        float4 *r_sh = a.r_shadow + 2 * (size_t)sg * a.seg;
        auto make_sink = [&](uint32_t pl, uint32_t lane) {
            RecordSink sink;
            sink.rec = a.rec; sink.s_rec_count = s_rec; sink.rec_cap = a.rec_cap;
            sink.film = a.film_out; sink.film_w = a.film.width; sink.bins = a.film.bins;
            sink.n_freq = a.film.n_freq; sink.freq = a.film.freq; sink.start_opl = a.film.start_opl;
            sink.p_local = pl; sink.p_seg = pl - pl0; sink.lane = lane;
            sink.n_splats = 0; sink.n_overflow = 0; sink.log = a.log;
            return sink;
        };
        if (DEFER && n_z) {      // zombies of the previous bounce: commit the parked term, count the path
            const uint32_t *qz = a.q_zombie + (size_t)par * a.n_slots + (size_t)sg * a.seg;
            for (uint32_t i = tid; i < ((n_z + 63u) & ~63u); i += kBlock) {
                const bool on = i < n_z;
                uint32_t pl = pl0;
                f3 L = mk(0, 0, 0);
                if (on) {
                    const uint32_t zq = qz[i], slot = sg * a.seg + (zq & 0xffffu);       // (shadow-list position << 16 | slot within the segment)
                    uint32_t pixel, s;
                    slot_to_lane(a, slot, pixel, s, pl);
                    if (a.occ[(size_t)sg * a.occ_stride + (zq >> 16)] == 0) {
                        const float4 pe = P.ld(Q_PEND, slot);
                        const uint32_t py = pixel / a.film.crop_w, px = pixel - a.film.crop_w * py;
                        RecordSink sink = make_sink(pl, pixel * a.rc.spp_total + s);
                        commit_pending(L, mk(pe.x, pe.y, pe.z), pe.w, P.ld_w(Q_AUX, slot) - 1u, px + a.film.crop_x, py + a.film.crop_y, a.film, a.rc, sink);
                        n_splats += sink.n_splats; n_over += sink.n_overflow;
                    }
                }
                wave_deposit(s_steady, on, pl - pl0, L, 1.0f);
            }
        }
        for (uint32_t k = 0; k < kWfKeys; ++k) {                 // one material type after the other
            const uint32_t n_k = a.seg_mat[(size_t)sg * kWfKeys + k];
            const uint32_t *q = a.q_mat + (size_t)k * a.n_slots + (size_t)sg * a.seg;
            const uint32_t n_round = (n_k + 63u) & ~63u;
            // the list entry of the NEXT pass is requested a pass ahead: entry -> state is a dependent pair of loads, and three waves per
            // SIMD do not hide two memory latencies per vertex
            uint32_t e_next = (uint32_t)tid < n_k ? q[tid] : 0u;
            for (uint32_t i = tid; i < n_round; i += kBlock) {
                const bool on = i < n_k;
                const uint32_t e = e_next;
                if (i + kBlock < n_k) e_next = q[i + kBlock];
                bool alive = false, zombie = false;
                uint32_t slot = 0, dep_px = 0, sh_pos = 0;
                f3 ray_o = mk(0, 0, 0), ray_d = mk(0, 0, 1), dep_L = mk(0, 0, 0); float ray_eta = 1.0f, dep_w = 0.0f;
                if (on) {
                    // e: (position in the live list the trace kernel walked, slot within the segment)
                    slot = sg * a.seg + (e & 0xffffu);
                    uint32_t pixel, s, pl;
                    slot_to_lane(a, slot, pixel, s, pl);
                    Path p;
                    uint32_t pend = 0u;
                    // (the path's live-list entry — its high half is where the shadow ray of the previous vertex sits in the shadow list —
                    // is requested with the state, not behind it)
                    const uint32_t live_entry = (DEFER && !FIRST) ? a.q_live[(size_t)par * a.n_slots + (size_t)sg * a.seg + (e >> 16)] : 0u;
                    if (FIRST) path_begin(p, a.cam, a.film, a.rc, pixel, s);      // (see k_wf_raygen: bounce 0's state is recomputed, not read)
                    else {
                        load_path(P, slot, r_cur[2 * (size_t)(e >> 16) + 1], p, pend);
                        const uint32_t py = pixel / a.film.crop_w, px = pixel - a.film.crop_w * py;
                        p.px = px + a.film.crop_x; p.py = py + a.film.crop_y; p.lane = pixel * a.rc.spp_total + s;
                        p.rng.inc = rng_inc_of(a.rc.seed, p.lane, a.rc.flags);
                    }
                    Hit h;
                    const uint32_t lpos = e >> 16;                   // this vertex's position in the list the trace kernel walked = where its hit record lives
                    { const float4 hq = P.ld(hit_plane(par), sg * a.seg + lpos); h.t = hq.x; h.u = hq.y; h.v = hq.z; h.prim = (int32_t)__float_as_uint(hq.w); }
                    if (!FIRST && h.prim >= 0 && (fbits(sv.tshade[h.prim].h[4].z) >> 16) != 0u) {
                        // an emitter was hit: its MIS weight wants the vertex the path came from = the previous bounce's hit (a live path's is
                        // valid), which lives at the path's position in the PREVIOUS list — the word of the ray list a tmax would take
                        const uint32_t prev_pos = __float_as_uint(r_cur[2 * (size_t)lpos].w);
                        const float4 hq = P.ld(hit_plane(par ^ 1u), sg * a.seg + prev_pos);
                        Hit hp; hp.t = hq.x; hp.u = hq.y; hp.v = hq.z; hp.prim = (int32_t)__float_as_uint(hq.w);
                        p.prev_p = hit_point(sv, hp);
                    }
                    ++n_closest;
                    RecordSink sink = make_sink(pl, p.lane);
                    // (the flag of the shadow ray this path emitted at its previous vertex: at that ray's position in the shadow list, which
                    // the path's live-list entry carries in its high half)
                    if (DEFER && pend && a.occ[(size_t)sg * a.occ_stride + (live_entry >> 16)] == 0) {          // the previous bounce's emitter sample was visible: commit it now
                        const float4 pe = P.ld(Q_PEND, slot);
                        commit_pending(p.L, mk(pe.x, pe.y, pe.z), pe.w, p.depth - 1u, p.px, p.py, a.film, a.rc, sink);
                    }
                    Pending pd; Ray shadow;
                    shadow.o = mk(0, 0, 0); shadow.d = mk(0, 0, 1); shadow.tmax = 0.0f;
                    if ((a.rc.flags & MTR_FLAG_CAMERA_UNWARP) && p.depth == 0u && h.prim >= 0) p.dist = -h.t;      // camera_unwarp: see k_wf_raygen
                    // scenes in HBM: nothing runs between shade_hit and shade_finish, the surface interaction is handed over instead of
                    // rebuilt from a second fetch of the shading record (config 5 at 256 spp 153.3 -> 152.4 ms; not with the extended
                    // shading code, whose registers are all taken: 170.1 -> 173.3 ms)
                    constexpr bool kKeepCtx = DEFER && !EXT;
                    HitCtx hc;
                    shade_hit<EXT, TR>(p, h, sv, a.film, a.rc, sink, pd, shadow, kKeepCtx ? &hc : nullptr);
                    bool occluded = false;
                    if (pd.has_shadow) {
                        ++n_shadow;
                        if (SCENE_LDS) {                  // short rays out of LDS: tracing them right here is cheaper (config 2: 168 vs 243 ms)
                            Hit sh = traverse<true, (TR & kTrLeafPair) != 0u, flat_kind(TR)>(sv, shadow.o, shadow.d, shadow.tmax, st);
                            occluded = sh.prim >= 0;
                        } else {
                            // the ray goes to the segment's shadow list (k_wf_trace, any-hit, runs next), the term is parked and
                            // withheld from shade_finish: `occluded` only gates the commit there
                            const uint32_t pos = wave_append(s_shadow_p, true);
                            sh_pos = pos;          // (the occlusion kernel needs no slot: its result goes to the ray's list position)
                            if (a.q_order_sh) s_sortkey_sh[pos] = (uint8_t)trace_sort_key(a, shadow.o, shadow.d);
                            r_sh[2 * (size_t)pos] = make_float4(shadow.o.x, shadow.o.y, shadow.o.z, shadow.tmax);
                            r_sh[2 * (size_t)pos + 1] = make_float4(shadow.d.x, shadow.d.y, shadow.d.z, 0.0f);
                            P.st(Q_PEND, slot, make_float4(pd.Lr.x, pd.Lr.y, pd.Lr.z, pd.opl));
                            occluded = true;
                        }
                    }
                    alive = shade_finish<EXT, TR>(p, h, occluded, pd, sv, a.film, a.rc, sink, kKeepCtx ? &hc : nullptr);
                    ++n_bounce;
                    n_splats += sink.n_splats; n_over += sink.n_overflow;
                    ray_o = p.ray.o; ray_d = p.ray.d; ray_eta = p.eta;
                    const uint32_t pend_now = (DEFER && pd.has_shadow) ? 1u : 0u;
                    // p.L is what THIS vertex added (the term committed above, emission, the emitter sample): it goes to the pixel's sum now
                    dep_px = pl - pl0; dep_L = p.L;
                    if (alive) { ++n_alive; store_path(P, slot, p, pend_now); }      // (a path that ended leaves nothing to read)
                    else if (pend_now) {                      // ended with a term parked: the next launch commits it and counts the path
                        zombie = true; ++n_alive;             // (the host's live count must keep the loop going for it)
                        P.st_w(Q_AUX, slot, p.depth);
                    }
                    else dep_w = 1.0f;
                }
                wave_deposit(s_steady, on & ((dep_L.x != 0.0f) | (dep_L.y != 0.0f) | (dep_L.z != 0.0f) | (dep_w != 0.0f)), dep_px, dep_L, dep_w);
                // wave64 stream compaction of the survivors into the segment's next live list
                if (__ballot(alive) != 0ull) {
                    const uint32_t pos = wave_append(s_next_p, alive);
                    if (alive) {
                        q_next[pos] = (sh_pos << 16) | (slot - sg * a.seg);
                        if (a.q_order) s_sortkey[pos] = (uint8_t)trace_sort_key(a, ray_o, ray_d);
                        // (.w: not the ray's tmax, which is infinite, but where this vertex's hit record lives — the next vertex's prev_pos)
                        r_next[2 * (size_t)pos] = make_float4(ray_o.x, ray_o.y, ray_o.z, __uint_as_float(e >> 16));
                        r_next[2 * (size_t)pos + 1] = make_float4(ray_d.x, ray_d.y, ray_d.z, ray_eta);
                    }
                }
                if (DEFER && __ballot(zombie) != 0ull) {
                    const uint32_t pos = wave_append(s_zombie_p, zombie);
                    if (zombie) qz_next[pos] = (sh_pos << 16) | (slot - sg * a.seg);
                }
            }
        }
        __syncthreads();
        if (a.q_order) trace_sort(s_sortkey, *s_next_p, s_hist, a.q_order + (size_t)sg * a.seg, tid);
        if (DEFER && a.q_order_sh) { __syncthreads(); trace_sort(s_sortkey_sh, *s_shadow_p, s_hist, a.q_order_sh + (size_t)sg * a.seg, tid); }
        if (tid == 0) { wf_segment_survivors(a, sg, *s_next_p, DEFER ? *s_zombie_p : 0u); if (DEFER) a.seg_shadow[sg] = *s_shadow_p; }
        for (uint32_t t = tid; t < npx; t += kBlock) a.rec_count[pl0 + t] = s_rec[t];
        for (uint32_t t = tid; t < 4 * npx; t += kBlock) {        // this workgroup owns the segment's pixels in this launch
            const float v = s_steady[t];
            if (v != 0.0f) {
                const uint32_t pixel = a.pix0 + pl0 + (t >> 2);
                const uint32_t cy = pixel / a.film.crop_w, cx = pixel - cy * a.film.crop_w;
                if (cx < a.film.width && cy < a.film.height) a.steady_out[((size_t)cy * a.film.width + cx) * 4u + (t & 3u)] += v;
            }
        }
        __syncthreads();
    }
    // counters: one set of atomics per wave (statistics only; `live_total` lets the host stop unbounded renders)
    if (a.counters) {
        const unsigned vals[5] = { n_closest, n_shadow, n_splats, n_bounce, n_over };
        unsigned long long *dst[5] = { &a.counters->rays_closest, &a.counters->rays_shadow, &a.counters->splats_issued,
                                       &a.counters->bounces, &a.counters->splats_overflow };
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            unsigned v = vals[k];
            for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
            if ((tid & 63) == 0 && v) atomicAdd(dst[k], (unsigned long long)v);
        }
    }
    if (a.live_total) {
        unsigned v = n_alive;
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
        if ((tid & 63) == 0 && v) atomicAdd(a.live_total, v);
    }
}

// ---- NLOS tier in the wavefront organisation ------------------------------------------------------
// One launch per bounce runs the WHOLE loop iteration of transient_nlos_path (nlos_bounce: closest hit, laser /
// emitter sampling with its shadow rays, hidden-geometry or BSDF sampling, Russian roulette) for every slot of the
// segment's live list; path state lives in the SoA planes between launches, contributions become per-pixel records for
// k_wf_scatter, survivors are compacted into the next live list.  It is the second, independent organisation of this tier
// (k_fused<NLOS> being the first): same per-path arithmetic (mtr_nlos.h), different machinery around it — state in HBM
// instead of registers, records + the stand-alone scatter-add instead of LDS row histograms, host loop over bounces.
template <int STACK, bool SCENE_LDS, bool EXT>
__global__ void __launch_bounds__(kBlock, 2) k_wf_nlos_bounce(const WfArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *s_next_p = (uint32_t *)smem;
    const int tid = threadIdx.x;
    SceneView sv; WStack<STACK> st; uint32_t off;
    wf_setup<STACK, SCENE_LDS>(a.sc, smem, tid, sv, st, off);
    uint32_t *s_rec = (uint32_t *)(smem + off);
    float *s_steady = (float *)(smem + off + al16(a.G * 4u));
    const PlanesT<!SCENE_LDS> P{ (float4 *)a.planes, a.n_slots };
    const uint32_t par = a.parity;
    uint32_t n_closest = 0, n_shadow = 0, n_bounce = 0, n_splats = 0, n_over = 0, n_alive = 0;
    wf_ticket_begin(a, tid);
    for (uint32_t sg = wf_next_segment(a, smem, tid, true); sg < a.n_seg; sg = wf_next_segment(a, smem, tid, false)) {
        const uint32_t pl0 = sg * a.G;
        const uint32_t npx = min(a.G, a.P - pl0);
        for (uint32_t t = tid; t < npx; t += kBlock) s_rec[t] = a.rec_count[pl0 + t];
        for (uint32_t t = tid; t < 4 * npx; t += kBlock) s_steady[t] = 0.0f;
        if (tid == 0) *s_next_p = 0u;
        __syncthreads();
        const uint32_t n_live = a.seg_live[(size_t)par * a.n_seg + sg];
        const uint32_t *q = a.q_live + (size_t)par * a.n_slots + (size_t)sg * a.seg;
        uint32_t *q_next = a.q_live + (size_t)(par ^ 1u) * a.n_slots + (size_t)sg * a.seg;
        const uint32_t n_round = (n_live + 63u) & ~63u;
        for (uint32_t i = tid; i < n_round; i += kBlock) {
            bool alive = false;
            uint32_t slot = 0;
            if (i < n_live) {
                slot = q[i];
                uint32_t pixel, s, pl;
                slot_to_lane(a, slot, pixel, s, pl);
                Path p;
                load_state(P, slot, p);
                const uint32_t py = pixel / a.film.crop_w, px = pixel - a.film.crop_w * py;
                p.px = px + a.film.crop_x; p.py = py + a.film.crop_y; p.lane = pixel * a.rc.spp_total + s;
                p.rng.inc = rng_inc_of(a.rc.seed, p.lane, a.rc.flags);
                RecordSink sink;
                sink.rec = a.rec; sink.s_rec_count = s_rec; sink.rec_cap = a.rec_cap;
                sink.film = a.film_out; sink.film_w = a.film.width; sink.bins = a.film.bins;
                sink.n_freq = 0u; sink.freq = nullptr; sink.start_opl = a.film.start_opl;
                sink.p_local = pl; sink.p_seg = pl - pl0; sink.lane = p.lane;
                sink.n_splats = 0; sink.n_overflow = 0; sink.log = a.log;
                BounceStats bs; bs.closest = 0; bs.shadow = 0;
                alive = nlos_bounce<EXT>(p, sv, a.nlos, a.film, a.rc, st, sink, bs);
                n_closest += bs.closest; n_shadow += bs.shadow; ++n_bounce;
                n_splats += sink.n_splats; n_over += sink.n_overflow;
                if (alive) { ++n_alive; store_state(P, slot, p, true); }
                else {
                    const uint32_t fx = p.px - a.film.crop_x, fy = p.py - a.film.crop_y;
                    if (fx < a.film.width && fy < a.film.height) {
                        float *sp = s_steady + 4 * (pl - pl0);
                        __hip_atomic_fetch_add(sp, p.L.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_fetch_add(sp + 1, p.L.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_fetch_add(sp + 2, p.L.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_fetch_add(sp + 3, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            }
            if (__ballot(alive) != 0ull) {
                const uint32_t pos = wave_append(s_next_p, alive);
                if (alive) q_next[pos] = slot;
            }
        }
        __syncthreads();
        if (tid == 0) wf_segment_survivors(a, sg, *s_next_p);
        for (uint32_t t = tid; t < npx; t += kBlock) a.rec_count[pl0 + t] = s_rec[t];
        for (uint32_t t = tid; t < 4 * npx; t += kBlock) {
            const float v = s_steady[t];
            if (v != 0.0f) {
                const uint32_t pixel = a.pix0 + pl0 + (t >> 2);
                const uint32_t cy = pixel / a.film.crop_w, cx = pixel - cy * a.film.crop_w;
                if (cx < a.film.width && cy < a.film.height) a.steady_out[((size_t)cy * a.film.width + cx) * 4u + (t & 3u)] += v;
            }
        }
        __syncthreads();
    }
    if (a.counters) {
        const unsigned vals[5] = { n_closest, n_shadow, n_splats, n_bounce, n_over };
        unsigned long long *dst[5] = { &a.counters->rays_closest, &a.counters->rays_shadow, &a.counters->splats_issued,
                                       &a.counters->bounces, &a.counters->splats_overflow };
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            unsigned v = vals[k];
            for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
            if ((tid & 63) == 0 && v) atomicAdd(dst[k], (unsigned long long)v);
        }
    }
    if (a.live_total) {
        unsigned v = n_alive;
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
        if ((tid & 63) == 0 && v) atomicAdd(a.live_total, v);
    }
}

// ---- the time-bin scatter-add: one workgroup per pixel of the tile ------------------------------
// LDS float atomics (ds_add_f32) retire at a fixed 3 clocks per LANE on gfx950 whatever the address
// pattern (204 G adds/s for the whole chip, measured), 64-bit integer LDS atomics at 2.7 T/s.  The row
// is therefore accumulated in signed 2^-42 fixed point with ds_add_u64 and converted back to
// f32 once per (pixel, bin) at the flush: 13x faster, order-independent (deterministic) sums, and an
// absolute rounding error of 2^-43 per contribution — below the f32 rounding of any bin sum > 1e-5.
// resolution 2.3e-13, range +-2^21 per bin.
// RANGE GUARD (round 6): the reference's film is plain f32 (transient_image_block.py:79-81: scatter_reduce(Add) — values of
// any size, Inf and NaN propagate).  A pixel's n records are summed in fixed point only while every channel value is below
// 2^20 / n in magnitude (then no bin sum can leave +-2^20); a value at or above that — or an Inf / a NaN, which fail the
// same comparison — marks the PIXEL, whose row is then rebuilt from its record stream with f32 LDS atomics, i.e. by the code
// of the f32-row instantiation (fixed_row_limit, the `redo` blocks of k_wf_scatter / k_splat_rows / k_splat_rows_rec).
__device__ __forceinline__ float fixed_row_limit(uint32_t n) { return 1048576.0f / (float)(n ? n : 1u); }
__device__ __forceinline__ bool fixed_row_unsafe(float r, float g, float b, float lim)
{
    return !(fabsf(r) < lim) || !(fabsf(g) < lim) || !(fabsf(b) < lim);
}
__device__ __forceinline__ unsigned long long to_fixed(float v)
{
    long long q = __float2ll_rn(v * 4398046511104.0f);          // 2^42, exact scaling
    if (q == 0 && v != 0.0f) q = v > 0.0f ? 1 : -1;               // never lose a contribution entirely
    return (unsigned long long)q;
}
__device__ __forceinline__ float from_fixed(unsigned long long h)
{
    return __ll2float_rn((long long)h) * 2.2737367544323206e-13f; // 2^-42
}

template <bool FIXED>
__global__ void __launch_bounds__(kBlock) k_wf_scatter(const WfArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *row = (float *)(smem + 64);                              // [3][T] f32 ...
    unsigned long long *row64 = (unsigned long long *)(smem + 64);  // ... or [3][T] fixed point
    const uint32_t T = a.film.bins;
    const int tid = threadIdx.x;
    const bool rows = a.rec_cap > 0;                    // false: T*12 B does not fit LDS, shade used HBM atomics
    uint32_t *s_redo = (uint32_t *)smem;                // (FIXED) RANGE GUARD: this pixel's row must be rebuilt in f32
    if (tid == 0) *s_redo = 0u;
    if (rows) {
        if (FIXED) for (uint32_t t = tid; t < 3 * T; t += kBlock) row64[t] = 0ull;
        else for (uint32_t t = tid; t < 3 * T; t += kBlock) row[t] = 0.0f;
    }
    __syncthreads();
    // The record stream of the NEXT pixel is requested before the current row is flushed.  Records and film rows are
    // touched exactly once: NON-TEMPORAL loads and stores keep them from displacing each other in L2 (config 2, 1.33 GB per
    // launch: 0.336 -> 0.281 ms = 50 -> 59 % of the HBM roof; the prefetch and a grid of one resident round of workgroups
    // alone changed nothing; without the LDS adds 0.314, without the film flush 0.239 ms)
    constexpr int kBatch = 8;                 // independent 16-byte loads in flight per lane (coalesced)
    uint4 r[kBatch];
    uint32_t n_next_all = blockIdx.x < a.P ? a.rec_count[blockIdx.x] : 0u;
    auto fetch = [&](uint32_t pl_, uint32_t n_, uint32_t base) {
        const uint4 *rec_ = a.rec + (size_t)pl_ * a.rec_cap;
#pragma unroll
        for (int k = 0; k < kBatch; ++k) {
            const uint32_t i = base + k * kBlock + tid;
            r[k] = (i < n_) ? nt_load(rec_ + i) : make_uint4(0xffffffffu, 0u, 0u, 0u);
        }
    };
    if (blockIdx.x < a.P) fetch(blockIdx.x, rows ? min(n_next_all, a.rec_cap) : 0u, 0u);
    for (uint32_t pl = blockIdx.x; pl < a.P; pl += gridDim.x) {
        const uint32_t pixel = a.pix0 + pl;
        const uint32_t cy = pixel / a.film.crop_w, cx = pixel - cy * a.film.crop_w;    // film coords (crop offset removed)
        const bool in_film = (cx < a.film.width) & (cy < a.film.height);
        const size_t fpix = (size_t)cy * a.film.width + cx;
        const uint32_t n_all = n_next_all;
        const uint32_t n = rows ? min(n_all, a.rec_cap) : 0u;
        const bool store_only = a.film_zero && n_all <= a.rec_cap;      // no overflow atomics landed on this row
        const uint32_t pl_next = pl + gridDim.x;
        if (pl_next < a.P) n_next_all = a.rec_count[pl_next];
        const float lim = fixed_row_limit(n);           // (FIXED) see RANGE GUARD above
        bool unsafe = false;
        for (uint32_t base = 0; base < n || base == 0u; base += kBatch * kBlock) {
            if (base != 0u) fetch(pl, n, base);
#pragma unroll
            for (int k = 0; k < kBatch; ++k) {
                if (r[k].x != 0xffffffffu) {
                    if (FIXED) {
                        unsigned long long *p = row64 + r[k].x;
                        unsafe |= fixed_row_unsafe(__uint_as_float(r[k].y), __uint_as_float(r[k].z), __uint_as_float(r[k].w), lim);
                        atomicAdd(p, to_fixed(__uint_as_float(r[k].y)));
                        atomicAdd(p + T, to_fixed(__uint_as_float(r[k].z)));
                        atomicAdd(p + 2 * T, to_fixed(__uint_as_float(r[k].w)));
                    } else {
                        float *p = row + r[k].x;
                        __hip_atomic_fetch_add(p, __uint_as_float(r[k].y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_fetch_add(p + T, __uint_as_float(r[k].z), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_fetch_add(p + 2 * T, __uint_as_float(r[k].w), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            }
        }
        if (pl_next < a.P) fetch(pl_next, rows ? min(n_next_all, a.rec_cap) : 0u, 0u);      // in flight across the flush below
        if (FIXED && unsafe) *s_redo = 1u;
        __syncthreads();
        bool as_f32 = !FIXED;
        if (FIXED && rows && *s_redo != 0u) {
            // RANGE GUARD: a value of this pixel does not fit the fixed-point row — the row again, in f32, from the record stream
            // (the next pixel's first batch stays in r[]: this block has its own loads)
            as_f32 = true;
            for (uint32_t t = tid; t < 6 * T; t += kBlock) row[t] = 0.0f;
            __syncthreads();
            if (tid == 0) *s_redo = 0u;
            const uint4 *rec_ = a.rec + (size_t)pl * a.rec_cap;
            for (uint32_t i = tid; i < n; i += kBlock) {
                const uint4 q = nt_load(rec_ + i);
                float *p = row + q.x;
                __hip_atomic_fetch_add(p, __uint_as_float(q.y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(p + T, __uint_as_float(q.z), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(p + 2 * T, __uint_as_float(q.w), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            __syncthreads();
        }
        if (rows) {
            float4 *dst = (float4 *)(a.film_out + fpix * T * 4u);
            for (uint32_t t = tid; t < T; t += kBlock) {
                float r_, g, b;
                bool nz;
                if (FIXED && !as_f32) {
                    const unsigned long long qr = row64[t], qg = row64[T + t], qb = row64[2 * T + t];
                    nz = (qr | qg | qb) != 0ull;
                    r_ = from_fixed(qr); g = from_fixed(qg); b = from_fixed(qb);
                    if (nz) { row64[t] = 0ull; row64[T + t] = 0ull; row64[2 * T + t] = 0ull; }
                } else {
                    r_ = row[t]; g = row[T + t]; b = row[2 * T + t];
                    nz = r_ != 0.0f || g != 0.0f || b != 0.0f;
                    if (nz) { row[t] = 0.0f; row[T + t] = 0.0f; row[2 * T + t] = 0.0f; }
                }
                if ((nz || store_only) && in_film) {          // store_only: whole lines, zeros included (the row is contiguous)
                    float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    if (!store_only) v = dst[t];                  // accumulate onto earlier passes / overflow atomics
                    v.x += r_; v.y += g; v.z += b;
                    nt_store(dst + t, v);
                }
            }
        }
        __syncthreads();
    }
}

// phasor_hdr_film: the frequency-domain counterpart of k_wf_scatter (phasor_image_block.py:42-67).  One workgroup per
// pixel; threads are arranged as (record lane) x (frequency): a chunk of the pixel's (opl, value) records is staged in
// LDS, every thread walks its share of the chunk for ITS frequency and keeps the complex sum in registers — no atomics
// per record (ds_add_f32 costs 3 clocks per lane) — and the record lanes are folded once per pixel.
constexpr uint32_t kPhasorChunk = 1024;
__global__ void __launch_bounds__(kBlock) k_wf_phasor_scatter(const WfArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2 *s_rec = (float2 *)(smem + 64);                          // [kPhasorChunk] (opl - start_opl, value)
    float2 *s_part = (float2 *)(smem + 64 + kPhasorChunk * 8u);     // [kBlock] partial sums of the threads
    const uint32_t F = a.film.n_freq;
    const int tid = threadIdx.x;
    for (uint32_t pl = blockIdx.x; pl < a.P; pl += gridDim.x) {
        const uint32_t pixel = a.pix0 + pl;
        const uint32_t cy = pixel / a.film.crop_w, cx = pixel - cy * a.film.crop_w;
        const bool in_film = (cx < a.film.width) & (cy < a.film.height);
        const uint32_t n = min(a.rec_count[pl], a.rec_cap);
        const uint4 *rec = a.rec + (size_t)pl * a.rec_cap;
        float *dst = a.film_out + ((size_t)cy * a.film.width + cx) * (2u * F + 1u);
        for (uint32_t f0 = 0; f0 < F; f0 += kBlock) {              // frequency chunks (one for F <= 256)
            const uint32_t fc = min(F - f0, (uint32_t)kBlock);
            const uint32_t lanes = kBlock / fc;                    // record lanes per frequency
            const uint32_t f = (uint32_t)tid % fc, rl = (uint32_t)tid / fc;
            const bool on = rl < lanes;
            const float freq = a.film.freq[f0 + f];
            float re = 0.0f, im = 0.0f;
            for (uint32_t base = 0; base < n; base += kPhasorChunk) {
                const uint32_t m = min(n - base, kPhasorChunk);
                __syncthreads();
                for (uint32_t i = tid; i < m; i += kBlock) {
                    const uint4 r = rec[base + i];
                    s_rec[i] = make_float2(__uint_as_float(r.x) - a.film.start_opl, __uint_as_float(r.y));   // phasor_hdr_film.py:249
                }
                __syncthreads();
                if (on)
                    for (uint32_t i = rl; i < m; i += lanes) {
                        const float2 r = s_rec[i];
                        float c, sn;
                        phasor_term(freq, r.x, c, sn);
                        re += r.y * c; im += r.y * sn;
                    }
            }
            __syncthreads();
            s_part[tid] = make_float2(re, im);
            __syncthreads();
            if ((uint32_t)tid < fc && in_film) {
                float sr = 0.0f, si = 0.0f;
                for (uint32_t l = 0; l < lanes; ++l) { const float2 v = s_part[l * fc + tid]; sr += v.x; si += v.y; }
                if (sr != 0.0f || si != 0.0f) { dst[2 * (f0 + tid)] += sr; dst[2 * (f0 + tid) + 1] += si; }
            }
        }
        __syncthreads();
    }
}

template <int STACK, bool SL>
hipError_t launch_set(const WfArgs &a, int which, int grid, size_t lds, hipStream_t stream)
{
    const bool ext = a.sc.has_rough != 0u;
    if constexpr (SL) {          // scenes staged in LDS whose tables allow it: the specialised shading code (as k_fused)
        if (which == 2 && !ext && (a.sc.traits & kTrCornell) == kTrCornell) {
            // (NOT the flat top level of k_fused: this kernel's lanes are the samples of neighbouring pixels at the SAME bounce — their shadow
            // rays walk the tree together — and flat_walk_device's uniform stages cost them more than the walk: config 2 in this
            // organisation 82.4 -> 87.4 ms with it, measured in round 6; a flat k_wf_trace changed nothing, 82.4 against 81 - 83)
            void (*ks)(const WfArgs) = a.first_bounce ? k_wf_shade<STACK, true, false, kTrCornell, true> : k_wf_shade<STACK, true, false, kTrCornell>;
            lds += al16(a.G * 4u) + al16(a.G * 16u) + al16(a.seg) + (a.q_order ? 1056u : 0u);       // (+ 260 words: trace_sort's histogram)
            hipError_t e = hipFuncSetAttribute((const void *)ks, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(ks, dim3(grid), dim3(kBlock), lds, stream, a);
            return hipGetLastError();
        }
    }
    void (*k)(const WfArgs) = which == 0 ? k_wf_raygen<STACK, SL> : which == 1 ? (a.trace_any ? k_wf_trace<STACK, SL, true> : a.first_bounce ? k_wf_trace<STACK, SL, false, true> : k_wf_trace<STACK, SL, false>)
                            : which == 5 ? (ext ? k_wf_nlos_bounce<STACK, SL, true> : k_wf_nlos_bounce<STACK, SL, false>)
                            : a.first_bounce ? (ext ? k_wf_shade<STACK, SL, true, 0u, true> : k_wf_shade<STACK, SL, false, 0u, true>)
                            : (ext ? k_wf_shade<STACK, SL, true> : k_wf_shade<STACK, SL, false>);
    lds += al16(a.G * 4u) + al16(a.G * 16u) + al16(a.seg);        // k_wf_shade: record-list tails, steady sums, sort keys of the next live list; k_wf_trace: hit material types
    if (which == 2 && a.q_order) lds += (SL ? 0u : al16(a.seg)) + 1056u;      // k_wf_shade, TRACE ORDER experiment: + sort keys of the shadow list (scenes in HBM), trace_sort's histogram
    hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(kBlock), lds, stream, a);
    return hipGetLastError();
}

} // namespace

size_t wf_planes_bytes(uint32_t n_slots) { return (size_t)PL16_COUNT * n_slots * 16u + (size_t)n_slots * 8u; }

bool wf_plan(const SceneDev &sc, WfConfig &cfg)
{
    if (sc.bvh_depth > 64) return false;
    cfg.stack = sc.bvh_depth <= 8 ? 8 : sc.bvh_depth <= 16 ? 16 : sc.bvh_depth <= 32 ? 32 : 64;
    uint32_t scene_b = al16(sc.n_wnodes * sizeof(WNode)) + al16(sc.n_slots / 2 * sizeof(TriPair)) + al16(sc.n_slots * sizeof(TriShade)) +
                       al16(sc.n_mats * sizeof(mtr_material)) + al16(sc.n_ems * sizeof(Emitter));
    cfg.scene_lds = sc.wnodes != nullptr && scene_b <= 64u * 1024u;
    cfg.lds_bytes = 64 + (size_t)wf_stack_rows(sc, cfg.scene_lds) * kBlock * 4 + (cfg.scene_lds ? scene_b : 0) + 16;
    return true;
}

// which: 0 raygen, 1 trace, 2 shade, 3 scatter, 5 NLOS bounce
hipError_t launch_wf(const WfArgs &a, const WfConfig &cfg, int which, int grid, hipStream_t stream)
{
    if (which == 3 && a.film.n_freq) {
        const size_t lds = 64 + kPhasorChunk * 8u + kBlock * 8u;
        hipLaunchKernelGGL(k_wf_phasor_scatter, dim3(grid), dim3(kBlock), lds, stream, a);
        return hipGetLastError();
    }
    if (which == 3) {
        // fixed-point rows need 24 B per bin; fall back to f32 rows (12 B) when that does not leave 2 workgroups per CU
        const bool fixed = a.rec_cap && (size_t)a.film.bins * 24u <= 72u * 1024u;
        size_t lds = 64 + (a.rec_cap ? (size_t)a.film.bins * (fixed ? 24u : 12u) : 16u);
        void (*k)(const WfArgs) = fixed ? k_wf_scatter<true> : k_wf_scatter<false>;
        hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        // one RESIDENT round of workgroups: with 24 KB rows six fit a CU; eight per CU would leave a second, third-full round
        // of workgroups behind the first
        int dev = 0, n_cu = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n_cu > 0) {
            int per_cu = (int)((160u * 1024u) / lds);
            if (per_cu > 8) per_cu = 8;
            if (per_cu < 1) per_cu = 1;
            if (grid > n_cu * per_cu) grid = n_cu * per_cu;
        }
        hipLaunchKernelGGL(k, dim3(grid), dim3(kBlock), lds, stream, a);
        return hipGetLastError();
    }
    // (the traversal stack is sized by the levels of the tree that is walked — wf_stack_rows — not by a compile-time class: ONE
    // instantiation of every kernel; rounds 1-4 compiled four identical copies, for stack classes 8 / 16 / 32 / 64)
    return cfg.scene_lds ? launch_set<64, true>(a, which, grid, cfg.lds_bytes, stream)
                         : launch_set<64, false>(a, which, grid, cfg.lds_bytes, stream);
}

} // namespace mtr
