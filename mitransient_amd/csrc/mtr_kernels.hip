// mtr_kernels.hip — CDNA4 (gfx950) kernels of the transient path tracer.
//
// k_fused        MTR_MODE_FUSED: one workgroup owns every gridDim-th pixel (all their samples), G of them at a time.
//                The whole scene (BVH2 node packets + triangles + materials) is staged in LDS,
//                paths are generated, traced and shaded by persistent lanes that refill
//                themselves from an LDS work counter (no idle lanes while the workgroup has work),
//                every OPL -> time-bin contribution is an LDS float atomic into the pixel's slot of a
//                private (G, T, 3) ring of row histograms, and each film row is added to HBM exactly
//                once, coalesced, by the wave that ended the pixel's last path: no global atomics, no
//                splat traffic, no workgroup barrier between pixels.
// k_splat_*      the stand-alone time-bin scatter-add (add_transient_data + put_ + accum).
// k_develop_*    TransientHDRFilm.develop.
#include "mtr_kernels.h"
#include "mtr_knobs.h"

#include <hip/hip_runtime.h>
#include <algorithm>
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

namespace mtr {

// ------------------------------------------------------------------ helpers
struct LdsStack {
    int32_t *base;     // &stack[tid]; entry k lives at base[k * kBlock]  (one bank column per lane)
    int sp;            // the column has FusedArgs::stack_rows rows: one per stacked group + the row push_if writes before it counts
    __device__ __forceinline__ void reset() { sp = 0; }
    // unconditional LDS write, conditional increment: no branch in the node step
    __device__ __forceinline__ void push_if(bool c, int32_t v) { base[sp * kBlock] = v; sp += c ? 1 : 0; }
    __device__ __forceinline__ int32_t pop() { --sp; return base[sp * kBlock]; }
    __device__ __forceinline__ bool empty() const { return sp == 0; }
#ifndef MTR_NO_PARK
    // PARKED PATH STATE: values a path needs once per bounce (or only when it hits an emitter) live in the lane's LDS column
    // below the stack rows instead of in registers across the two traversals of a bounce: prev_p (3 dwords), the PCG32
    // increment (2), prev_pdf (1).  path_bounce reads them back where they are used.  Config 2, same box: 69.30 -> 68.65 ms,
    // L2 requests -10 %, scratch written back to HBM -10 % (the kernel needs 161 registers and runs with 128).
    static constexpr bool kPark = true;
    static constexpr uint32_t kParkRows = 6;
    int park_row;      // first parking row = the kernel's stack_rows
    __device__ __forceinline__ void park_prev_p(f3 v) { base[(park_row + 0) * kBlock] = (int32_t)fbits(v.x); base[(park_row + 1) * kBlock] = (int32_t)fbits(v.y); base[(park_row + 2) * kBlock] = (int32_t)fbits(v.z); }
    __device__ __forceinline__ f3 unpark_prev_p() const { return mk(bitsf((uint32_t)base[(park_row + 0) * kBlock]), bitsf((uint32_t)base[(park_row + 1) * kBlock]), bitsf((uint32_t)base[(park_row + 2) * kBlock])); }
    __device__ __forceinline__ void park_inc(uint64_t v) { base[(park_row + 3) * kBlock] = (int32_t)(uint32_t)v; base[(park_row + 4) * kBlock] = (int32_t)(uint32_t)(v >> 32); }
    __device__ __forceinline__ uint64_t unpark_inc() const { return (uint64_t)(uint32_t)base[(park_row + 3) * kBlock] | ((uint64_t)(uint32_t)base[(park_row + 4) * kBlock] << 32); }
    __device__ __forceinline__ void park_prev_pdf(float v) { base[(park_row + 5) * kBlock] = (int32_t)fbits(v); }
    __device__ __forceinline__ float unpark_prev_pdf() const { return bitsf((uint32_t)base[(park_row + 5) * kBlock]); }
#else
    static constexpr bool kPark = false;
    static constexpr uint32_t kParkRows = 0;
    __device__ __forceinline__ void park_prev_p(f3) {}
    __device__ __forceinline__ f3 unpark_prev_p() const { return mk(0, 0, 0); }
    __device__ __forceinline__ void park_inc(uint64_t) {}
    __device__ __forceinline__ uint64_t unpark_inc() const { return 0; }
    __device__ __forceinline__ void park_prev_pdf(float) {}
    __device__ __forceinline__ float unpark_prev_pdf() const { return 0.0f; }
#endif
#ifdef MTR_PROFILE_CYCLES      // experiment build: wave-clock per code section (time since the previous mark)
    unsigned long long t0, cyc[6];
#if MTR_PROFILE_CYCLES == 2     // ... the flat walk's stages in sections 0 (box selection), 5 (box faces), 2 (rectangle slabs), 4 (rectangle tests); everything else but shading in 3
    __device__ __forceinline__ void prof_mark(int sec) { unsigned long long t = __builtin_readcyclecounter(); cyc[sec == 1 ? 1 : 3] += t - t0; t0 = t; }
    __device__ __forceinline__ void prof_flat(int sec) { unsigned long long t = __builtin_readcyclecounter(); cyc[sec] += t - t0; t0 = t; }
#else
    __device__ __forceinline__ void prof_mark(int sec) { unsigned long long t = __builtin_readcyclecounter(); cyc[sec] += t - t0; t0 = t; }
    __device__ __forceinline__ void prof_flat(int) {}
#endif
#else
    __device__ __forceinline__ void prof_mark(int) {}
    __device__ __forceinline__ void prof_flat(int) {}
#endif
#ifdef MTR_PROFILE_SIMT        // experiment build: lane-steps vs wave-steps of node / triangle tests
    unsigned int ls[2], ws[2], wmax, wcalls;
    // wave maximum of the per-lane node-step count of one traverse() call (the floor of its wave-steps)
    __device__ __forceinline__ void tail(unsigned int mine) {
        unsigned int mx = 0;
        for (int bit = 11; bit >= 0; --bit) { const unsigned int c = mx | (1u << bit); if (__ballot(mine >= c) != 0ull) mx = c; }
        unsigned long long m = __ballot(1);
        if ((int)(threadIdx.x & 63) == __ffsll((long long)m) - 1) { wmax += mx; wcalls += 1; }
    }
    __device__ __forceinline__ void count(int k) {
        ls[k]++;
        unsigned long long m = __ballot(1);
        if ((int)(threadIdx.x & 63) == __ffsll((long long)m) - 1) ws[k]++;
    }
#else
    __device__ __forceinline__ void count(int) {}
#endif
};

__device__ __forceinline__ void lds_add(float *p, float v)
{
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__device__ __forceinline__ void log_splat(const SplatLog &lg, uint32_t lane, uint32_t depth, uint32_t kind,
                                          uint32_t pixel, uint32_t bin, float r, float g, float b, float opl)
{
    unsigned long long i = atomicAdd(lg.count, 1ull);
    if (i < lg.cap) {
        uint32_t *R = lg.rec + 8 * i;
        R[0] = lane; R[1] = depth | (kind << 16); R[2] = pixel; R[3] = bin;
        R[4] = __float_as_uint(r); R[5] = __float_as_uint(g); R[6] = __float_as_uint(b); R[7] = __float_as_uint(opl);
    }
}

// signed 2^-42 fixed point for order-independent LDS sums (ds_add_u64): resolution 2.3e-13, range +-2^21; a non-zero
// contribution never rounds to zero
__device__ __forceinline__ unsigned long long splat_to_fixed(float v)
{
    long long q = __float2ll_rn(v * 4398046511104.0f);          // 2^42
    if (q == 0 && v != 0.0f) q = v > 0.0f ? 1 : -1;
    return (unsigned long long)q;
}
__device__ __forceinline__ float splat_from_fixed(unsigned long long q) { return __ll2float_rn((long long)q) * 2.2737367544323206e-13f; }
// RANGE GUARD of the fixed-point rows (round 6; the reference's film is plain f32: transient_image_block.py:79-81).  n values are
// summed in fixed point only while each is below 2^20 / n in magnitude — no sum then leaves +-2^20; a larger value, an Inf or a
// NaN (they fail the same comparison) takes the f32 route of the kernel it is in (see mtr_wavefront.hip: to_fixed).
__device__ __forceinline__ float splat_fixed_limit(uint64_t n) { return 1048576.0f / (float)(n ? n : 1ull); }
__device__ __forceinline__ bool splat_fixed_unsafe(float r, float g, float b, float lim)
{
    return !(fabsf(r) < lim) || !(fabsf(g) < lim) || !(fabsf(b) < lim);
}

#ifndef MTR_FUSED_SEG_LANES
#define MTR_FUSED_SEG_LANES 2048u      // samples in flight per workgroup the ring of row slots is sized for (1024 / 4096 measured worse)
#endif

// the workgroup's ring of row histograms in LDS: planes [3][G*T]
template <bool GREY = false>          // GREY (scene trait kTrGrey): r == g == b in every contribution, the row holds ONE plane
struct LdsHistSink {
    float *hist; uint32_t plane;       // plane = G * T
    uint32_t row;                      // (local pixel) * T, set per path
    uint32_t film_w;
    uint32_t lane;
    uint32_t n_splats;
    SplatLog log;
    __device__ __forceinline__ void splat(uint32_t fx, uint32_t fy, uint32_t bin, float r, float g, float b,
                                          float opl, uint32_t depth, uint32_t kind)
    {
        float *p = hist + row + bin;
        if (GREY) lds_add(p, r);
        else { lds_add(p, r); lds_add(p + plane, g); lds_add(p + 2 * plane, b); }
        ++n_splats;
        if (log.rec) log_splat(log, lane, depth, kind, fy * film_w + fx, bin, r, g, b, opl);
    }
};

// MTR_FLAG_DETERMINISTIC: the same ring in 64-bit fixed point.  RANGE GUARD: a path adds at most two contributions to any one bin
// per depth, so the contributions of depth < dcap whose channels are all below lim = 2^20 / (2 spp dcap) cannot carry a bin
// sum out of +-2^20; every other one (deeper, larger, Inf, NaN) goes to the f32 OVERFLOW ring beside the fixed-point one
// (ovf: [3][G * T] f32) and the flush returns fixed + overflow — what the reference's f32 film holds, up to summation order.
struct LdsFixedSink {
    unsigned long long *hist; float *ovf; uint32_t plane, row, film_w, lane, n_splats;
    float lim; uint32_t dcap;
    SplatLog log;
    __device__ __forceinline__ void splat(uint32_t fx, uint32_t fy, uint32_t bin, float r, float g, float b,
                                          float opl, uint32_t depth, uint32_t kind)
    {
        if (depth < dcap && !splat_fixed_unsafe(r, g, b, lim)) {
            unsigned long long *p = hist + row + bin;
            atomicAdd(p, splat_to_fixed(r)); atomicAdd(p + plane, splat_to_fixed(g)); atomicAdd(p + 2 * plane, splat_to_fixed(b));
        } else {
            float *o = ovf + row + bin;
            lds_add(o, r); lds_add(o + plane, g); lds_add(o + 2 * plane, b);
        }
        ++n_splats;
        if (log.rec) log_splat(log, lane, depth, kind, fy * film_w + fx, bin, r, g, b, opl);
    }
};

// phasor_hdr_film in the fused kernel: the workgroup's ring holds (Re, Im) per frequency instead of time bins —
// [G][2F] — and every contribution adds its F terms (phasor_image_block.py:42-67) with LDS float atomics
struct LdsPhasorSink {
    float *row;                        // the pixel's slot: 2F floats
    const float *freq; uint32_t n_freq; float start_opl;
    uint32_t film_w, lane, n_splats;
    SplatLog log;
    __device__ __forceinline__ void splat(uint32_t fx, uint32_t fy, uint32_t bin, float r, float g, float b,
                                          float opl, uint32_t depth, uint32_t kind)
    {
        const float rel = opl - start_opl;                                   // phasor_hdr_film.py:249
        for (uint32_t f = 0; f < n_freq; ++f) {
            float c, sn;
            phasor_term(freq[f], rel, c, sn);
            lds_add(row + 2u * f, r * c); lds_add(row + 2u * f + 1u, r * sn);
        }
        ++n_splats;
        if (log.rec) log_splat(log, lane, depth, kind, fy * film_w + fx, bin, r, g, b, opl);
    }
};

// contract form: f32 atomics straight into the (H,W,T,4) tensor in HBM
struct GlobalAtomicSink {
    float *film; uint32_t film_w, bins;
    uint32_t lane;
    uint32_t n_splats;
    SplatLog log;
    __device__ __forceinline__ void splat(uint32_t fx, uint32_t fy, uint32_t bin, float r, float g, float b,
                                          float opl, uint32_t depth, uint32_t kind)
    {
        size_t idx = (((size_t)fy * film_w + fx) * bins + bin) * 4u;
        unsafeAtomicAdd(film + idx, r); unsafeAtomicAdd(film + idx + 1, g); unsafeAtomicAdd(film + idx + 2, b);
        ++n_splats;
        if (log.rec) log_splat(log, lane, depth, kind, fy * film_w + fx, bin, r, g, b, opl);
    }
};

__device__ __forceinline__ void copy16(void *dst, const void *src, uint32_t bytes, int tid)
{
    const uint4 *s = (const uint4 *)src; uint4 *d = (uint4 *)dst;
    for (uint32_t i = tid; i < bytes / 16u; i += kBlock) d[i] = s[i];
}
__host__ __device__ constexpr uint32_t align16(uint32_t x) { return (x + 15u) & ~15u; }

// ------------------------------------------------------------------ fused kernel
// the run of FusedArgs that the path start and the end-of-path bookkeeping read (kernarg_copy)
struct LoopArgs { uint32_t spp_begin, spp_chunk, G; FastDiv div_spp, div_G; };
static_assert(offsetof(FusedArgs, spp_chunk) - offsetof(FusedArgs, spp_begin) == offsetof(LoopArgs, spp_chunk) &&
              offsetof(FusedArgs, G) - offsetof(FusedArgs, spp_begin) == offsetof(LoopArgs, G) &&
              offsetof(FusedArgs, div_spp) - offsetof(FusedArgs, spp_begin) == offsetof(LoopArgs, div_spp) &&
              offsetof(FusedArgs, div_G) - offsetof(FusedArgs, spp_begin) == offsetof(LoopArgs, div_G), "LoopArgs mirrors FusedArgs");
#ifndef MTR_FUSED_MIN_WAVES
#define MTR_FUSED_MIN_WAVES 4          // waves per SIMD the register allocator must leave room for
#endif
// MINW: waves per SIMD the register allocator must leave room for.  4 when four workgroups fit a CU; long rows (one
// 48 KB histogram per workgroup: three per CU) get the 168-register budget of 3 waves per SIMD instead of spilling.
// PHASOR: phasor_hdr_film (HIST_LDS form only): the ring rows hold (Re, Im) per frequency, the flush adds 2F floats per pixel
// FIXED: MTR_FLAG_DETERMINISTIC (HIST_LDS form only): rows and steady sums in 64-bit fixed point
// TR: scene traits (mtr_core.h: kTrDiffuse | kTrOneRectEmitter) — shading code the scene's tables cannot reach is not compiled in
template <bool SCENE_LDS, bool HIST_LDS, bool NLOS, int MINW = MTR_FUSED_MIN_WAVES, bool PHASOR = false, bool FIXED = false, bool ROUGH = false, uint32_t TR = 0u>
__global__ void __launch_bounds__(kBlock, MINW) k_fused(const FusedArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
#ifdef MTR_PROFILE_TAIL
    const unsigned long long t0_wall = wall_clock64();
#endif

    // ---- LDS carve-up (all offsets multiples of 16) ----
    uint32_t off = 0;
    int32_t *s_stack = (int32_t *)(smem + off); off += (a.stack_rows + (NLOS ? 0u : LdsStack::kParkRows)) * kBlock * 4;      // (the NLOS loop parks nothing)
    unsigned long long *s_cnt = (unsigned long long *)(smem + off); off += 64;    // 5 counters + next
    uint32_t *s_next = (uint32_t *)(s_cnt + 6);
    SceneView sv;
    sv.n_emitters = a.sc.n_ems; sv.n_slots = a.sc.n_slots;
    sv.samp_tris = a.sc.samp_tris; sv.samp_vn = a.sc.samp_vn; sv.face_pmf = a.sc.face_pmf; sv.face_cdf = a.sc.face_cdf; sv.vnormals = a.sc.vnormals;
    sv.texels = a.sc.texels; sv.tex_info = a.sc.tex_info; sv.uvs = a.sc.uvs;
    sv.flat_off = (uint32_t)(offsetof(FusedArgs, sc) + offsetof(SceneDev, flat));
    if (SCENE_LDS) {
        // a scene staged in LDS is walked through its 8-wide tree (fused_plan); the BVH2 packets stay in HBM, unused
        const uint32_t tree_bytes = a.sc.n_wnodes * (uint32_t)sizeof(WNode);
        WNode *n = (WNode *)(smem + off); off += align16(tree_bytes);
        TriPair *tg = (TriPair *)(smem + off); off += align16(a.sc.n_slots / 2 * sizeof(TriPair));
        TriShade *ts = (TriShade *)(smem + off); off += align16(a.sc.n_slots * sizeof(TriShade));
        mtr_material *mm = (mtr_material *)(smem + off); off += align16(a.sc.n_mats * sizeof(mtr_material));
        Emitter *ee = (Emitter *)(smem + off); off += align16(a.sc.n_ems * sizeof(Emitter));
        copy16(n, a.sc.wnodes, align16(tree_bytes), tid);
        copy16(tg, a.sc.tpairs, align16(a.sc.n_slots / 2 * sizeof(TriPair)), tid);
        copy16(ts, a.sc.tshade, align16(a.sc.n_slots * sizeof(TriShade)), tid);
        copy16(mm, a.sc.mats, align16(a.sc.n_mats * sizeof(mtr_material)), tid);
        copy16(ee, a.sc.ems, align16(a.sc.n_ems * sizeof(Emitter)), tid);
        sv.nodes = nullptr; sv.wnodes = n; sv.wnodes4 = nullptr; sv.wnodes8q = nullptr; sv.tpairs = tg; sv.tshade = ts; sv.mats = mm; sv.ems = ee;
        sv.node_pairs = true;
        __builtin_assume(sv.wnodes != nullptr);          // (an LDS address: the compiler does not know it cannot be null, and would keep the other walkers of traverse())
    } else {
        sv.nodes = a.sc.nodes; sv.tpairs = a.sc.tpairs; sv.tshade = a.sc.tshade; sv.mats = a.sc.mats; sv.ems = a.sc.ems;
        sv.wnodes = nullptr; sv.wnodes4 = a.sc.wnodes4; sv.wnodes8q = a.sc.wnodes8q;
        sv.node_pairs = false;
    }
    // ---- work distribution: workgroups draw CHUNKS of a.chunk consecutive pixels from a global ticket counter (pixels
    // differ in cost by more than 10x — the Cornell box's image has empty margins — and a static assignment left the
    // cheapest workgroup of config 2 idle after 6 of 102 ms; rotating the assignment still left a 52 .. 90 ms spread).
    // ---- pixel ring inside a chunk: K = a.G row slots; the chunk's q-th pixel lives in slot q % K from its first sample
    // until its last path has ended, then the wave that ended it flushes the row and hands the slot to pixel q + K:
    // lanes of pixel q + 1 start while pixel q drains; the only workgroup barrier is the one between chunks.
    const uint32_t K = a.G;
    float *s_steady = (float *)(smem + off); off += align16(K * 32);          // [K][4] f32, or u64 fixed point (FIXED)
    unsigned long long *s_steady64 = (unsigned long long *)s_steady;
    float *s_steady_ovf = (float *)(smem + off); if (FIXED) off += align16(K * 16);      // (FIXED) [K][4] f32: RANGE GUARD overflow of the steady sums
    uint32_t *s_owner = (uint32_t *)(smem + off); off += align16(K * 4);      // pixel ordinal that may use the slot
    uint32_t *s_done = (uint32_t *)(smem + off); off += align16(K * 4);       // paths of that pixel that have ended
    float *s_hist = (float *)(smem + off);
    const uint32_t T = PHASOR ? 2u * a.film.n_freq : a.film.bins;       // floats of one plane of a row
    // (kTrGrey: one plane per row — the "three" planes of the flush below are the same one, read three times)
    constexpr bool kGrey = (TR & kTrGrey) != 0u;
    const uint32_t plane = kGrey ? 0u : K * T;

    if (tid < 6) s_cnt[tid] = 0ull;
    for (uint32_t k = tid; k < K; k += kBlock) s_done[k] = 0u;
    for (uint32_t k = tid; k < K * 8; k += kBlock) s_steady[k] = 0.0f;
    unsigned long long *s_hist64 = (unsigned long long *)s_hist;
    float *s_ovf = s_hist + 6u * plane;                 // (FIXED) the f32 overflow ring behind the fixed-point one: [3][plane]
    if (HIST_LDS) for (uint32_t k = tid; k < (kGrey ? K * T : (PHASOR ? 1u : (FIXED ? 9u : 3u)) * plane); k += kBlock) s_hist[k] = 0.0f;
    if (FIXED) for (uint32_t k = tid; k < K * 4; k += kBlock) s_steady_ovf[k] = 0.0f;

    LdsStack st; st.base = s_stack + tid; st.sp = 0;
#ifndef MTR_NO_PARK
    st.park_row = (int)a.stack_rows;
#endif
#ifdef MTR_PROFILE_SIMT
    st.ls[0] = st.ls[1] = st.ws[0] = st.ws[1] = 0; st.wmax = 0; st.wcalls = 0;
#endif
#ifdef MTR_PROFILE_CYCLES
    for (int k = 0; k < 6; ++k) st.cyc[k] = 0;
    st.t0 = __builtin_readcyclecounter();
#endif
    // Counters are kept per WAVE, summed from ballots (scalar registers, scalar adds): five per-lane counters would be
    // five vector registers live across every traversal of a kernel that already spills.  The NLOS loop can trace many
    // shadow rays per bounce (one per illuminated point), so that variant counts per lane.
    unsigned long long w_closest = 0, w_shadow = 0, w_bounce = 0, w_paths = 0, w_splats = 0;
#ifdef MTR_PROFILE_OCC
    unsigned long long occ_iter = 0, occ_alive = 0, occ_wait = 0;
#endif
    uint32_t n_closest = 0, n_shadow = 0, n_splats = 0;          // NLOS only

    uint32_t *s_chunk = s_next + 1;                  // [2]: first pixel of the ticket, pixels in it
    const uint32_t n_px_all = a.pixel_end - a.pixel_begin;
    if (tid == 0) { s_chunk[0] = 0u; s_chunk[1] = 0u; }        // (no chunk left yet: the band accounting below reads the PREVIOUS ticket)
    for (;;) {
    // every wave has left the previous chunk (all its rows are flushed): draw the next one, reset the ring
    __syncthreads();
    if (tid == 0 && a.n_bands && s_chunk[1] != 0u && s_chunk[0] < n_px_all) {
        // BAND COMPLETION WORDS (mtr_render_params.n_bands).  Every row of the chunk this workgroup has just left is flushed (the
        // barrier above: workgroup-scope release of every wave's stores).  ONE agent-scope release for the chunk, then its pixels
        // are added to the count of the band(s) they belong to; whoever completes a band publishes the epoch at system scope: a
        // stream parked on that word (hipStreamWaitValue32: the band's film reduction) proceeds while this launch renders on.
        // (A fence per flushed pixel instead: 60.2 against 59.4 ms on config 2; per chunk: see tools/bands.py.)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        uint32_t f = s_chunk[0];
        const uint32_t end = f + min(s_chunk[1], n_px_all - f);
        while (f < end) {
            // equal bands of band_px = floor(n / n_bands) pixels, the LAST takes the remainder (no band is ever empty: a word
            // nobody would publish would park its waiter for ever)
            const uint32_t bnd = min(f / a.band_px, a.n_bands - 1u);
            const bool last = bnd == a.n_bands - 1u;
            const uint32_t lim = last ? end : min(end, (bnd + 1u) * a.band_px), cnt = lim - f;
            const uint32_t in_band = last ? n_px_all - bnd * a.band_px : a.band_px;
            const uint32_t old = __hip_atomic_fetch_add(a.band_count + bnd, cnt, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (old + cnt == in_band) __hip_atomic_store(a.band_done + bnd, a.band_epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            f = lim;
        }
    }
    if (tid == 0) {
        // guided: a.chunk pixels per ticket, fewer as the launch runs out (about half a share of what is left), so that the
        // workgroups finish within one pixel of each other — a launch per row band (multi-GPU pipeline) ends 8 times per render
        const uint32_t seen = __hip_atomic_load(a.ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t left = seen < n_px_all ? n_px_all - seen : 0u;
        uint32_t want = left / (2u * gridDim.x);
        want = want < 1u ? 1u : (want > a.chunk ? a.chunk : want);
        s_chunk[0] = atomicAdd(a.ticket, want); s_chunk[1] = want;
        *s_next = kBlock;
    }
    for (uint32_t k = tid; k < K; k += kBlock) s_owner[k] = k;
    __syncthreads();
    const uint32_t first = s_chunk[0];
    if (first >= n_px_all) break;
    const uint32_t pix0 = a.pixel_begin + first;
    const uint32_t n_lanes = min(s_chunk[1], n_px_all - first) * a.spp_chunk;
    auto pixel_of = [&](uint32_t qq) -> uint32_t { return pix0 + qq; };
    st.prof_mark(5);

    // ---- persistent lanes: refill from the LDS work counter when a path ends ----
    // The __ballot()s are convergent operations: they pin this loop to ONE wave-synchronous
    // iteration = (start waiting lanes) + (one bounce for every live lane) + (flush finished rows).  Without
    // them the compiler threads `alive` through the back edge and nests a per-path inner loop, i.e. dead
    // lanes wait for the longest path of their wave instead of refilling (measured: 1/3 of the time).
    // (A finer-grained wave scheduler — node / leaf / shade blocks picked by lane-count ballots —
    // was measured too: 2.6x SLOWER; see DESIGN.md "what did not work".)
    uint32_t i = tid;
    bool alive = false;
    bool waiting = i < n_lanes;          // holds a sample index whose path has not started yet
    Path p;
    p.L = mk(0, 0, 0);
    uint32_t q = 0, slot = 0;
    uint32_t xy = 0;                     // (packed film coordinates, below) px | py << 16 of the lane's path
    bool carry = false;                  // the path that just ended leaves its radiance in p.L: this lane's next path is in the same pixel
    for (;;) {
        st.prof_mark(4);                 // (experiment builds) sections: 0 traversal, 1 shading, 2 path start, 3 end-of-path bookkeeping, 4 row flush, 5 idle
        if (__ballot(waiting) != 0ull) {
            bool started = false;
            // (the arguments a path start needs: read from the kernarg segment here, see kernarg_copy)
            const LoopArgs la = kernarg_copy<LoopArgs>(offsetof(FusedArgs, spp_begin));
            if (waiting) {
                q = fastdiv(i, la.div_spp);
                slot = q - fastdiv(q, la.div_G) * la.G;
                if (__hip_atomic_load(s_owner + slot, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == q) {
                    const uint32_t s = la.spp_begin + (i - q * la.spp_chunk);
                    const uint32_t pixel = pixel_of(q);
                    const f3 L_carried = carry ? p.L : mk(0, 0, 0);
                    if (NLOS) {
                        const NlosConst nc_s = kernarg_copy<NlosConst>(offsetof(FusedArgs, nlos));
                        const Film film_s = kernarg_copy<Film>(offsetof(FusedArgs, film));
                        const RenderConst rc_s = kernarg_copy<RenderConst>(offsetof(FusedArgs, rc));
                        nlos_begin(p, nc_s, film_s, rc_s, pixel, s);
                    }
                    else {
                        // COLD KERNEL ARGUMENTS FROM THE KERNARG SEGMENT (round 5).  The camera (34 dwords) is read only here; as a by-value
                        // argument it lived in scalar registers across the whole persistent loop, i.e. in v_readlane'd spill slots
                        // (the loop holds more than 100 uniform values).  Scalar loads from the kernarg segment where a path starts
                        // instead: 111 -> 82 spilled SGPRs, config 2 61.1 -> 60.4 ms (same box)
                        const Camera cam_l = kernarg_copy<Camera>(offsetof(FusedArgs, cam));
                        const Film film_s = kernarg_copy<Film>(offsetof(FusedArgs, film));
                        const RenderConst rc_s = kernarg_copy<RenderConst>(offsetof(FusedArgs, rc));
                        path_begin(p, cam_l, film_s, rc_s, pixel, s);
                        if (LdsStack::kPark) { st.park_inc(p.rng.inc); st.park_prev_p(p.prev_p); st.park_prev_pdf(p.prev_pdf); }
                    }
                    p.L = L_carried; carry = false;
                    xy = p.px | (p.py << 16);
                    alive = true; waiting = false; started = true;
                }
            }
            const uint32_t n_started = (uint32_t)__popcll(__ballot(started));
            w_paths += n_started;
            if (!NLOS && (a.rc.flags & MTR_FLAG_CAMERA_UNWARP)) w_closest += n_started;
        }
        st.prof_mark(2);
        if (__ballot(alive) == 0ull) {
            if (__ballot(waiting) == 0ull) break;        // the whole wave is out of work
            __builtin_amdgcn_s_sleep(8);                 // every lane waits for a row another wave is about to flush
            st.prof_mark(5);
            continue;
        }
        bool closes = false;
#ifdef MTR_PROFILE_OCC         // experiment build: occupancy of the persistent lanes (wave iterations, alive lanes, waiting lanes)
        occ_iter += 1; occ_alive += (uint32_t)__popcll(__ballot(alive)); occ_wait += (uint32_t)__popcll(__ballot(waiting));
#endif
        w_bounce += (uint32_t)__popcll(__ballot(alive));
        uint32_t did_shadow = 0u, did_splats = 0u;                 // this iteration's per-lane counts (0..1, 0..2)
        if (alive) {
            BounceStats bstat; bstat.closest = 0; bstat.shadow = 0;
            // PACKED FILM COORDINATES (round 4).  A path's film coordinates px, py are read after either traversal (the in-film test
            // of its contributions) and at its end, and nowhere inside a traversal: they ride in ONE register (xy), which
            // path_bounce's `refresh` hook unpacks after either traversal — the compiler cannot keep the unpacked pair alive
            // instead, because xy passes through an empty asm each time.  Together with the unused tree walkers folded out of
            // traverse() (the assume on sv.wnodes above) this takes config 2's scratch from 112 to 84 B per lane — under what the L2
            // slices hold for the resident waves: WRITE_SIZE per launch 23.5 -> 5.8 GB (3.2 GB of it the film) — at 65.0 ms.
            // (Measured beside it, same box: everything recomputed from the sample index — pixel ordinal, row slot, coordinates,
            // lane id: 68 B, 3.5 GB, but 66.1 ms; nothing packed: 96 B, 64.3 ms, 14.7 GB; xy + the row slot from q: 76 B, 65.3 ms,
            // 4.0 GB.  DESIGN.md section 6.)
            // camera_unwarp: no traversal of its own (a third inlined copy of the walk in every instantiation) — bounce 0's closest
            // hit is the camera ray's; the extra ray the reference traces is still COUNTED (w_closest, at path start)
            const bool unwarp = !NLOS && (a.rc.flags & MTR_FLAG_CAMERA_UNWARP) != 0u;
            // ... and the film / render constants (29 dwords) come from the kernarg segment after either traversal, where the
            // shading reads them, instead of staying in scalar registers (spill slots) through the walks (kernarg_copy, above)
            Film film_l = a.film; RenderConst rc_l = a.rc;
            // the NLOS loop (four walks per bounce, ~110 dwords of projector / wall / table constants): the same, after every walk
            NlosConst nc_l = a.nlos;
            auto reload_nlos = [&]() {
                nc_l = kernarg_copy<NlosConst>(offsetof(FusedArgs, nlos));
                film_l = kernarg_copy<Film>(offsetof(FusedArgs, film)); rc_l = kernarg_copy<RenderConst>(offsetof(FusedArgs, rc));
            };
            // (NOT the splat log and NOT the arguments of the end-of-path bookkeeping: re-read where they are used as well, 37 instead
            // of 65 spilled SGPRs but 59.0 -> 59.9 / 59.6 ms — short sections wait for their scalar loads)
            auto refresh = [&](Path &pp, auto &) {
                uint32_t w = xy; asm volatile("" : "+v"(w));
                pp.px = w & 0xffffu; pp.py = w >> 16;
                if (!NLOS) { film_l = kernarg_copy<Film>(offsetof(FusedArgs, film)); rc_l = kernarg_copy<RenderConst>(offsetof(FusedArgs, rc)); }
            };
            if (PHASOR) {
                LdsPhasorSink sink; sink.row = s_hist + slot * T; sink.freq = a.film.freq; sink.n_freq = a.film.n_freq;
                sink.start_opl = a.film.start_opl; sink.film_w = a.film.width; sink.lane = p.lane; sink.n_splats = 0; sink.log = a.log;
                alive = path_bounce<ROUGH, TR>(p, sv, film_l, rc_l, st, sink, bstat, refresh, unwarp);
                did_splats = sink.n_splats;
            } else if (FIXED) {
                LdsFixedSink sink; sink.hist = s_hist64; sink.ovf = s_ovf; sink.plane = plane; sink.row = slot * T;
                sink.lim = a.fixed_lim; sink.dcap = a.fixed_dcap;
                sink.film_w = a.film.width; sink.lane = p.lane; sink.n_splats = 0; sink.log = a.log;
                alive = NLOS ? nlos_bounce<ROUGH, TR>(p, sv, nc_l, film_l, rc_l, st, sink, bstat, reload_nlos)
                             : path_bounce<ROUGH, TR>(p, sv, film_l, rc_l, st, sink, bstat, refresh, unwarp);
                if (NLOS) n_splats += sink.n_splats; else did_splats = sink.n_splats;
            } else if (HIST_LDS) {
                LdsHistSink<(TR & kTrGrey) != 0u> sink; sink.hist = s_hist; sink.plane = plane; sink.row = slot * T;
                sink.film_w = a.film.width; sink.lane = p.lane; sink.n_splats = 0; sink.log = a.log;
                alive = NLOS ? nlos_bounce<ROUGH, TR>(p, sv, nc_l, film_l, rc_l, st, sink, bstat, reload_nlos)
                             : path_bounce<ROUGH, TR>(p, sv, film_l, rc_l, st, sink, bstat, refresh, unwarp);
                if (NLOS) n_splats += sink.n_splats; else did_splats = sink.n_splats;
            } else {
                GlobalAtomicSink sink; sink.film = a.film_out; sink.film_w = a.film.width; sink.bins = T;
                sink.lane = p.lane; sink.n_splats = 0; sink.log = a.log;
                alive = NLOS ? nlos_bounce<ROUGH, TR>(p, sv, nc_l, film_l, rc_l, st, sink, bstat, reload_nlos)
                             : path_bounce<ROUGH, TR>(p, sv, film_l, rc_l, st, sink, bstat, refresh, unwarp);
                if (NLOS) n_splats += sink.n_splats; else did_splats = sink.n_splats;
            }
            if (NLOS) { n_closest += bstat.closest; n_shadow += bstat.shadow; } else did_shadow = bstat.shadow;
            if (!NLOS && !alive) { uint32_t w = xy; asm volatile("" : "+v"(w)); p.px = w & 0xffffu; p.py = w >> 16; }      // (packed film coordinates, above)
            if (!alive) {
                // steady splat: block.put(pos, [L.r, L.g, L.b, 1])  (common.py:187-200).  The lane takes its next sample first:
                // when that is another sample of the SAME pixel (with 1024 spp a lane runs four in a row), the radiance simply
                // keeps accumulating in p.L and is deposited with the last of them — three LDS float atomics per RUN of paths
                // instead of per path (they retire at 3 clocks per lane on gfx950).  The pixel cannot close in between: this
                // lane's next path of it has not ended.  Not for the order-independent rows (the runs depend on timing).
                // The sample COUNT needs no atomic at all: the row is flushed when all spp_chunk paths of the pixel have ended
                // (config 2: 68.7 -> 67.8 ms for that one atomic per path; fixed-point sums instead of f32: 67.7 -> 67.95, not kept).
                i = atomicAdd(s_next, 1u);
                waiting = i < n_lanes;
                carry = !FIXED && waiting && (i - q * a.spp_chunk) < a.spp_chunk;      // the next sample is still in pixel q (i only grows)
                const uint32_t fx = p.px - a.film.crop_x, fy = p.py - a.film.crop_y;
                if (fx < a.film.width && fy < a.film.height && !carry) {
                    if (FIXED) {
                        // (RANGE GUARD: spp_chunk radiances below 2^20 / spp_chunk cannot leave +-2^20; the others are summed in f32)
                        if (!splat_fixed_unsafe(p.L.x, p.L.y, p.L.z, a.fixed_lim * (float)(2u * a.fixed_dcap))) {
                            unsigned long long *sp = s_steady64 + 4 * slot;
                            atomicAdd(sp, splat_to_fixed(p.L.x)); atomicAdd(sp + 1, splat_to_fixed(p.L.y)); atomicAdd(sp + 2, splat_to_fixed(p.L.z));
                        } else {
                            float *sp = s_steady_ovf + 4 * slot;
                            lds_add(sp, p.L.x); lds_add(sp + 1, p.L.y); lds_add(sp + 2, p.L.z);
                        }
                    } else {
                        float *sp = s_steady + 4 * slot;
                        lds_add(sp, p.L.x); lds_add(sp + 1, p.L.y); lds_add(sp + 2, p.L.z);
                    }
                }
                // acq_rel: this lane's row / steady adds are performed before the count that may release the row.
                // (One atomic per wave for the next samples and one per (wave, slot) for the count — ballots, readlanes and a
                // leader lane — was measured: end-of-path bookkeeping 8.0 -> 11.0 % of the wave's time; not kept.)
                closes = __hip_atomic_fetch_add(s_done + slot, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP) + 1u == a.spp_chunk;
            }
        }
        if (!NLOS) {          // one closest-hit ray per live lane (counted with w_bounce), at most one shadow ray, at most two contributions
            w_shadow += (uint32_t)__popcll(__ballot(did_shadow != 0u));
            w_splats += (uint32_t)__popcll(__ballot((did_splats & 1u) != 0u)) + 2u * (uint32_t)__popcll(__ballot((did_splats & 2u) != 0u));
        }
        st.prof_mark(3);
        // ---- flush: each film row is touched once, by the wave that ended its last path, coalesced (16 B / lane) ----
        for (unsigned long long cm = __ballot(closes); cm != 0ull; cm &= cm - 1ull) {
            const int src = __ffsll((long long)cm) - 1;
            const uint32_t fq = __builtin_amdgcn_readlane(q, src), fs = __builtin_amdgcn_readlane(slot, src);
            const uint32_t pixel = pixel_of(fq);
            const uint32_t cy = pixel / a.film.crop_w, cx = pixel - cy * a.film.crop_w;   // == film coords
            const uint32_t wl = tid & 63u;
            if (cx < a.film.width && cy < a.film.height) {
                const size_t fpix = (size_t)cy * a.film.width + cx;
                if (PHASOR) {          // (H, W, 2F + 1): Re, Im per frequency, then the weight channel (stays 0)
                    float *dst = a.film_out + fpix * (size_t)(T + 1u);
                    float *h = s_hist + fs * T;
                    for (uint32_t t = wl; t < T; t += 64u) {
                        const float v = h[t];
                        if (v != 0.0f) { dst[t] = (a.rc.flags & MTR_FLAG_FILM_ZERO) ? v : dst[t] + v; h[t] = 0.0f; }
                    }
                } else if (HIST_LDS && (a.rc.flags & MTR_FLAG_DEVELOPED_ROWS)) {
                    // the caller's tensor is the DEVELOPED one, (H, W, T, 3): this launch holds every sample of the pixel, the
                    // weight channel is identically 0 (develop divides by 1), so the row goes out whole — zeros included,
                    // contiguous 12 bytes per lane — and neither a cleared film nor a develop pass is needed.  Written once,
                    // never read here: non-temporal, so the stream does not displace the waves' scratch lines from L2.
                    float *row3 = a.film_out + fpix * (size_t)T * 3u;
                    for (uint32_t t = wl; t < T; t += 64u) {      // (four bins per lane and pass, 16-byte accesses: no faster — 68.7 vs 68.7 ms — and two more spills)
                        float r, gc, b;
                        if (FIXED) {
                            unsigned long long *h = s_hist64 + fs * T;
                            float *o = s_ovf + fs * T;
                            r = splat_from_fixed(h[t]) + o[t]; gc = splat_from_fixed(h[t + plane]) + o[t + plane]; b = splat_from_fixed(h[t + 2 * plane]) + o[t + 2 * plane];
                            h[t] = 0ull; h[t + plane] = 0ull; h[t + 2 * plane] = 0ull;
                            o[t] = 0.0f; o[t + plane] = 0.0f; o[t + 2 * plane] = 0.0f;
                        } else {
                            float *h = s_hist + fs * T;
                            r = h[t]; gc = h[t + plane]; b = h[t + 2 * plane];
                            h[t] = 0.0f; h[t + plane] = 0.0f; h[t + 2 * plane] = 0.0f;
                        }
                        float *o = row3 + 3u * t;
                        __builtin_nontemporal_store(r, o); __builtin_nontemporal_store(gc, o + 1); __builtin_nontemporal_store(b, o + 2);
                    }
                } else if (FIXED) {
                    float4 *row = (float4 *)(a.film_out + fpix * T * 4u);
                    unsigned long long *h = s_hist64 + fs * T;
                    float *o = s_ovf + fs * T;
                    for (uint32_t t = wl; t < T; t += 64u) {
                        const unsigned long long qr = h[t], qg = h[t + plane], qb = h[t + 2 * plane];
                        const float er = o[t], eg = o[t + plane], eb = o[t + 2 * plane];       // (RANGE GUARD overflow; NaN != 0)
                        if ((qr | qg | qb) != 0ull || er != 0.0f || eg != 0.0f || eb != 0.0f) {
                            float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                            if (!(a.rc.flags & MTR_FLAG_FILM_ZERO)) v = row[t];
                            v.x += splat_from_fixed(qr) + er; v.y += splat_from_fixed(qg) + eg; v.z += splat_from_fixed(qb) + eb;
                            row[t] = v;
                            h[t] = 0ull; h[t + plane] = 0ull; h[t + 2 * plane] = 0ull;
                            o[t] = 0.0f; o[t + plane] = 0.0f; o[t + 2 * plane] = 0.0f;
                        }
                    }
                } else if (HIST_LDS) {
                    float4 *row = (float4 *)(a.film_out + fpix * T * 4u);
                    float *h = s_hist + fs * T;
                    // four bins per lane and pass (16-byte LDS reads, 64 contiguous bytes of film per lane); rows and planes
                    // are 16-byte aligned when T is a multiple of 4, the tail (or an odd T) goes bin by bin
                    // (only in the 3-waves instantiation, whose rows are long: in the 128-register one the extra live values
                    // of this block shift spills into the traversal loop, config 2 +5 %)
                    const uint32_t T4 = (MINW < 4 && (T & 3u) == 0u) ? T : 0u;
                    for (uint32_t t = 4u * wl; t < T4; t += 256u) {
                        const float4 r4 = *(const float4 *)(h + t), g4 = *(const float4 *)(h + t + plane), b4 = *(const float4 *)(h + t + 2 * plane);
                        const float rr[4] = { r4.x, r4.y, r4.z, r4.w }, gg[4] = { g4.x, g4.y, g4.z, g4.w }, bb[4] = { b4.x, b4.y, b4.z, b4.w };
                        bool any = false;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            if (rr[k] != 0.0f || gg[k] != 0.0f || bb[k] != 0.0f) {
                                float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                                if (!(a.rc.flags & MTR_FLAG_FILM_ZERO)) v = row[t + k];     // accumulate onto earlier passes
                                v.x += rr[k]; v.y += gg[k]; v.z += bb[k];
                                row[t + k] = v;
                                any = true;
                            }
                        }
                        if (any) {
                            const float4 z = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                            *(float4 *)(h + t) = z; *(float4 *)(h + t + plane) = z; *(float4 *)(h + t + 2 * plane) = z;
                        }
                    }
                    for (uint32_t t = T4 + wl; t < T; t += 64u) {
                        float r = h[t], gc = h[t + plane], b = h[t + 2 * plane];
                        if (r != 0.0f || gc != 0.0f || b != 0.0f) {
                            float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                            if (!(a.rc.flags & MTR_FLAG_FILM_ZERO)) v = row[t];     // accumulate onto earlier passes
                            v.x += r; v.y += gc; v.z += b;
                            row[t] = v;
                            h[t] = 0; h[t + plane] = 0; h[t + 2 * plane] = 0;
                        }
                    }
                }
                if (wl < 4) {
                    float v;
                    if (FIXED) { v = splat_from_fixed(s_steady64[4 * fs + wl]) + s_steady_ovf[4 * fs + wl]; s_steady64[4 * fs + wl] = 0ull; s_steady_ovf[4 * fs + wl] = 0.0f; }
                    else { v = s_steady[4 * fs + wl]; s_steady[4 * fs + wl] = 0.0f; }
                    if (wl == 3) v = (float)a.spp_chunk;          // every sample of the pixel of this launch has ended
                    if (v != 0.0f) a.steady_out[fpix * 4u + wl] += v;
                }
            }
            if (wl == 0) {
                s_done[fs] = 0u;
                __hip_atomic_store(s_owner + fs, fq + K, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    }       // chunks
    __syncthreads();

    // ---- counters: LDS reduction, then one set of global atomics per workgroup ----
    if (NLOS) {
        atomicAdd(&s_cnt[1], (unsigned long long)n_closest);
        atomicAdd(&s_cnt[2], (unsigned long long)n_shadow);
        atomicAdd(&s_cnt[3], (unsigned long long)n_splats);
    } else w_closest += w_bounce;
    if ((tid & 63) == 0) {
        atomicAdd(&s_cnt[0], w_paths);
        atomicAdd(&s_cnt[1], w_closest);
        atomicAdd(&s_cnt[2], w_shadow);
        atomicAdd(&s_cnt[3], w_splats);
        atomicAdd(&s_cnt[4], w_bounce);
    }
    __syncthreads();
    if (tid < 5 && a.counters) atomicAdd(&a.counters->paths + tid, s_cnt[tid]);
#ifdef MTR_PROFILE_OCC
    if ((tid & 63) == 0 && a.counters) {
        atomicAdd(&a.counters->splats_overflow, occ_iter); atomicAdd(&a.counters->r0, occ_alive); atomicAdd(&a.counters->r1, occ_wait);
    }
#endif
#ifdef MTR_PROFILE_TAIL       // experiment build: when does the first / the last workgroup finish (100 MHz wall clock)
    if (tid == 0 && a.counters) {
        const unsigned long long t = wall_clock64();
        atomicMax(&a.counters->r0, ~(t - t0_wall));           // shortest workgroup (counters start at 0: minimum kept as the maximum of the complement)
        atomicMax(&a.counters->r1, t - t0_wall);              // longest workgroup
        atomicAdd(&a.counters->splats_overflow, t - t0_wall); // sum over workgroups
    }
#endif
#ifdef MTR_PROFILE_SIMT
    if (a.counters) {
        unsigned long long *c = &a.counters->splats_overflow;
        atomicAdd(c + 0, ((unsigned long long)st.ls[0] << 32) | st.ws[0]);      // node: lane-steps | wave-steps... (per thread, summed)
        atomicAdd(c + 1, ((unsigned long long)st.ls[1] << 32) | st.ws[1]);
        atomicAdd(c + 2, ((unsigned long long)st.wcalls << 32) | (unsigned long long)st.wmax);     // traverse() calls of the wave | sum of their wave-maxima
    }
#endif
#ifdef MTR_PROFILE_CYCLES
    if ((tid & 63) == 0 && a.counters) {      // one lane per wave; sections: see the top of the persistent loop
        st.prof_mark(5);
        unsigned long long *c = &a.counters->splats_overflow;      // reuse 3 spare u64 slots: packed pairs of 32-bit Mcycles
        atomicAdd(c + 0, ((st.cyc[0] >> 10) << 32) | (st.cyc[1] >> 10));
        atomicAdd(c + 1, ((st.cyc[2] >> 10) << 32) | (st.cyc[3] >> 10));
        atomicAdd(c + 2, ((st.cyc[4] >> 10) << 32) | (st.cyc[5] >> 10));
    }
#endif
}

static uint32_t scene_lds_bytes(const SceneDev &sc)
{
    return align16(sc.n_wnodes * sizeof(WNode)) + align16(sc.n_slots / 2 * sizeof(TriPair)) +
           align16(sc.n_slots * sizeof(TriShade)) + align16(sc.n_mats * sizeof(mtr_material)) +
           align16(sc.n_ems * sizeof(Emitter));
}

bool fused_plan(const SceneDev &sc, const Film &film, uint32_t n_pixels, uint32_t spp_chunk, int n_cu,
                FusedArgs &args, FusedConfig &cfg)
{
    const uint32_t kLdsMax = 160u * 1024u;
    int stack = sc.bvh_depth <= 8 ? 8 : sc.bvh_depth <= 16 ? 16 : sc.bvh_depth <= 32 ? 32 : 64;
    if (sc.bvh_depth > 64) return false;
    // (k_fused keeps a path's film coordinates packed in one register, 16 bits each)
    if ((uint64_t)film.crop_x + film.crop_w > 65536ull || (uint64_t)film.crop_y + film.crop_h > 65536ull) return false;
    uint32_t scene_b = scene_lds_bytes(sc);
    cfg.scene_lds = sc.wnodes != nullptr && scene_b <= 64u * 1024u;
    // the kernel walks a WIDE tree (8-wide in LDS, quantised 4-wide in HBM): one stacked group per level (+ the row the
    // branch-free push writes before it knows whether it counts)
    const uint32_t rows = (cfg.scene_lds ? sc.wide_levels : (sc.wnodes8q ? sc.wide8q_levels : sc.wide4_levels)) + 1u;
    args.stack_rows = rows;
    uint32_t fixed_b = (rows + (args.nlos_on ? 0u : LdsStack::kParkRows)) * kBlock * 4 + 64;
    if (cfg.scene_lds) fixed_b += scene_b;
    // row slots: enough lanes in flight to keep 256 persistent threads busy, rows must fit in LDS
    const bool det = (args.rc.flags & MTR_FLAG_DETERMINISTIC) && !film.n_freq;
    // kTrGrey (NLOS loop, scene in LDS, f32 rows, the shading code without lobes — the instantiations launch_fused_s holds): ONE plane per row
    cfg.rough = sc.has_rough != 0u;
    const bool grey = args.nlos_on && !film.n_freq && !det && cfg.scene_lds && (sc.traits & kTrGrey) && (!cfg.rough || (sc.traits & kTrNoLobes)) &&
                      film.bins * 4u <= 48u * 1024u && !mtr::knob("MTR_NO_GREY");
    const uint32_t row_bytes = film.n_freq ? film.n_freq * 8u : film.bins * (det ? 36u : grey ? 4u : 12u);     // (Re, Im) per frequency | 3 planes of T bins (f32 | 64-bit fixed point + its f32 overflow ring) | one plane
    cfg.fixed = false;
    const uint32_t hist_budget = (det ? 108u : 48u) * 1024u;
    // RANGE GUARD of the fixed-point rows (LdsFixedSink): depth cap and per-channel limit of what is summed in fixed point
    args.fixed_dcap = args.rc.max_depth < 64u ? (args.rc.max_depth ? args.rc.max_depth : 1u) : 64u;
    args.fixed_lim = 1048576.0f / (2.0f * (float)(spp_chunk ? spp_chunk : 1u) * (float)args.fixed_dcap);
    uint32_t g_want = (MTR_FUSED_SEG_LANES + spp_chunk - 1u) / (spp_chunk ? spp_chunk : 1u);
    if (g_want < 1) g_want = 1;
    uint32_t g_fit = row_bytes ? hist_budget / row_bytes : 1u;
    cfg.hist_lds = true;
    uint32_t G;
    if (g_fit >= 1) G = g_want < g_fit ? g_want : g_fit;
    else if (fixed_b + row_bytes + 64 <= kLdsMax) G = 1;                  // one long row still fits the CU
    else { G = g_want; cfg.hist_lds = false; }                          // row > LDS: f32 atomics to HBM
    cfg.fixed = det && cfg.hist_lds;
    cfg.traits = (cfg.rough ? (sc.traits & kTrNoLobes) : (sc.traits & ~kTrNoLobes)) & ~kTrGrey;
    if (grey) cfg.traits |= kTrGrey;                 // (g_fit >= 1 by the bound on the bins above: hist_lds holds)
    // the kernel with the flat top level (launch_fused_s picks it under exactly this condition) walks no tree: no stack rows
    if (!args.nlos_on && !film.n_freq && !cfg.rough && cfg.scene_lds && cfg.hist_lds &&
        (cfg.fixed ? cfg.traits == kTrCornellFlat : (cfg.traits & kTrFlatFlags) == kTrFlatFlags)) {
        fixed_b -= rows * kBlock * 4;
        args.stack_rows = 0u;
    }
    // ROW SLOTS AGAINST WORKGROUPS PER CU (round 6).  A workgroup is one wave per SIMD; how many are resident is bounded by LDS (rows) and by
    // registers: the kernels launch_fused_s picks hold 168 registers (three workgroups per CU: the extended shading, the fixed-point rows, the
    // NLOS loop — whose 128-register form is 18 % SLOWER at four per CU than the 168-register one at three, measured) or 128 (four).  Residency is
    // worth more than the ring: config 4's share with (slots, workgroups per CU) = (1, 3) 6.23 ms, (2, 3) 5.94, (3, 3) 5.96, (2, 2) 7.89;
    // deterministic config 2 at 256 spp (3, 1) 47.3 ms against (1, 3) 27.9; 2048 bins (2, 2) 23.8 against (1, 4) 20.8 ms.  So: the slot count
    // that fits the most workgroups a CU can hold, and among those the largest (rounds 1 - 5 took as many slots as the row budget held).
    const int reg_cap = (cfg.rough || cfg.fixed || args.nlos_on) ? 3 : (int)MTR_FUSED_MIN_WAVES;
    auto lds_of = [&](uint32_t g) { return (size_t)fixed_b + align16(g * 32) + (cfg.fixed ? align16(g * 16) : 0u) + 2 * align16(g * 4) + (cfg.hist_lds ? (size_t)g * row_bytes : 0) + 16; };
    auto per_cu_of = [&](uint32_t g) { const int p = (int)(kLdsMax / lds_of(g)); return p > reg_cap ? reg_cap : p; };
    if (cfg.hist_lds && !mtr::knob("MTR_FUSED_OLD_PLAN")) {
        uint32_t best = G;
        for (uint32_t g = G; g-- > 1u;) if (per_cu_of(g) > per_cu_of(best)) best = g;
        G = best;
    }
    if (const char *e = mtr::knob("MTR_FUSED_G")) { const uint32_t v = (uint32_t)atoi(e); if (v >= 1u && v < G && cfg.hist_lds) G = v; }     // experiments
    if (G > n_pixels) G = n_pixels ? n_pixels : 1;
    if (G > 4096) G = 4096;
    args.G = G;
    args.div_G = fastdiv_make(G); args.div_spp = fastdiv_make(spp_chunk);
    cfg.stack = stack;
    cfg.lds_bytes = lds_of(G);
    if (cfg.lds_bytes > kLdsMax) return false;
    // persistent grid: as many workgroups as can be resident, each with at least g_want pixels
    int per_cu = per_cu_of(G);
    if (per_cu < 1) per_cu = 1;
    if (mtr::knob("MTR_FUSED_OLD_PLAN")) { per_cu = (int)(kLdsMax / cfg.lds_bytes); if (per_cu > 8) per_cu = 8; if (cfg.rough && per_cu > 3) per_cu = 3; }
    if (const char *e = mtr::knob("MTR_FUSED_PER_CU")) { const int v = atoi(e); if (v >= 1 && v < per_cu) per_cu = v; }     // experiments
    if (mtr::knob("MTR_FUSED_VERBOSE")) fprintf(stderr, "fused_plan: G %u, row %u B, lds %zu B, per_cu %d, traits %u, rough %d, fixed %d\n", G, row_bytes, (size_t)cfg.lds_bytes, per_cu, cfg.traits, (int)cfg.rough, (int)cfg.fixed);
    cfg.per_cu = per_cu;
    long grid = (long)n_cu * per_cu;
    uint32_t g_blk = g_want;                       // small renders: rather more workgroups than long pixel queues
    while (g_blk > 1 && ((long)n_pixels + g_blk - 1) / g_blk < n_cu && (unsigned long long)g_blk * spp_chunk > 512ull) g_blk = (g_blk + 1) / 2;
    const long max_blocks = ((long)n_pixels + g_blk - 1) / g_blk;
    if (grid > max_blocks) grid = max_blocks;
    if (grid < 1) grid = 1;
    // chunk of consecutive pixels per ticket: about 32768 samples (amortises the drain at the chunk's end: config 2 with
    // 4 / 8 / 16 / 32 / 64 pixels per ticket 72.8 / 71.7 / 71.2 / 70.9 / 70.9 ms), but at least 2 chunks per workgroup (the guided tickets
    // of the kernel — fewer pixels as the launch runs out — level the workgroups out; rounds 1 - 5 asked for 8 chunks per workgroup, which cut
    // the tickets of a 64-row band of config 2 to 4 pixels: eight band launches 55.9 ms, with 16-pixel tickets 54.9; config 4's share
    // 5.955 -> 5.92 ms with 42 instead of 10 pixels per ticket)
    uint32_t chunk = (32768u + spp_chunk - 1u) / (spp_chunk ? spp_chunk : 1u);
    const uint32_t c_bal = (uint32_t)((unsigned long long)n_pixels / (2ull * (unsigned long long)grid));
    if (chunk > c_bal) chunk = c_bal;
    if (chunk < 1u) chunk = 1u;
    if (const char *e = mtr::knob("MTR_FUSED_CHUNK")) { const int v = atoi(e); if (v >= 1) chunk = (uint32_t)v; }     // experiments
    if ((unsigned long long)chunk * spp_chunk > 0xffff0000ull) return false;     // the per-chunk sample counter is 32 bits wide
    args.chunk = chunk;
    args.n_chunks = (n_pixels + chunk - 1u) / chunk;
    if (grid > (long)args.n_chunks) grid = args.n_chunks;
    cfg.grid = (int)grid;
    return true;
}

#ifndef MTR_C2_TRAITS
#define MTR_C2_TRAITS kTrCornellFlat       // (-DMTR_ONLY_C2 builds) the instantiation config 2 runs; kTrCornell with MTR_NO_FLAT=1
#endif
template <bool NLOS>
static hipError_t launch_fused_s(const FusedArgs &args, const FusedConfig &cfg, hipStream_t stream)
{
    void (*k)(const FusedArgs) = nullptr;
#ifdef MTR_ONLY_C2            // tools/regs_c2.sh, tools/build_variant.sh -DMTR_ONLY_C2: compile ONLY the instantiation config 2 runs (experiments: one
                              // minute per variant instead of four); such a library renders config 2 and refuses everything else
    if (NLOS || args.film.n_freq || cfg.rough || cfg.fixed || !cfg.scene_lds || !cfg.hist_lds || cfg.traits != MTR_C2_TRAITS || cfg.per_cu <= 3) return hipErrorInvalidValue;
    k = k_fused<true, true, false, MTR_FUSED_MIN_WAVES, false, false, false, MTR_C2_TRAITS>;
#else
    if (!NLOS && args.film.n_freq) {
        if (!cfg.hist_lds || cfg.rough) return hipErrorInvalidValue;       // (2F floats per row always fit: fused_plan)
        k = cfg.scene_lds ? k_fused<true, true, false, MTR_FUSED_MIN_WAVES, true> : k_fused<false, true, false, MTR_FUSED_MIN_WAVES, true>;
    }
    else if (cfg.rough) {                  // scenes with GGX lobes / smooth normals / bitmaps: the f32 organisations only, 168 registers for the larger shading
        if (cfg.fixed) return hipErrorInvalidValue;
        if (NLOS && (cfg.traits & kTrGrey)) { if constexpr (NLOS) k = k_fused<true, true, true, 3, false, false, true, kTrNoLobes | kTrGrey>; }      // ... and one plane per row (fused_plan)
        else if (cfg.scene_lds && cfg.hist_lds && (cfg.traits & kTrNoLobes)) k = k_fused<true, true, NLOS, 3, false, false, true, kTrNoLobes>;      // ... for normals / bitmaps only: no lobe code
        else
        k = cfg.scene_lds ? (cfg.hist_lds ? k_fused<true, true, NLOS, 3, false, false, true> : k_fused<true, false, NLOS, 3, false, false, true>)
                          : (cfg.hist_lds ? k_fused<false, true, NLOS, 3, false, false, true> : k_fused<false, false, NLOS, 3, false, false, true>);
    }
    else if (cfg.fixed) {
        k = cfg.scene_lds ? k_fused<true, true, NLOS, 3, false, true> : k_fused<false, true, NLOS, 3, false, true>;
        // deterministic rows over a Cornell-class scene: the specialised shading code and the flat top level as well
        if constexpr (!NLOS) if (cfg.scene_lds && cfg.traits == kTrCornellFlat) k = k_fused<true, true, false, 3, false, true, false, kTrCornellFlat>;
    }
    else if (!NLOS && cfg.scene_lds && cfg.hist_lds && cfg.traits == kTrCornellFlat)      // ... and a flat top level: no tree walk either (flat_walk_device)
        k = cfg.per_cu <= 3 ? k_fused<true, true, false, 3, false, false, false, kTrCornellFlat> : k_fused<true, true, false, MTR_FUSED_MIN_WAVES, false, false, false, kTrCornellFlat>;
    else if (!NLOS && cfg.scene_lds && cfg.hist_lds && (cfg.traits & kTrFlatGeneral) == kTrFlatFlags)    // the flat top level under the general shading code (a mirror box, two lights ...)
        k = cfg.per_cu <= 3 ? k_fused<true, true, false, 3, false, false, false, kTrFlatFlags> : k_fused<true, true, false, MTR_FUSED_MIN_WAVES, false, false, false, kTrFlatFlags>;
    else if (!NLOS && cfg.scene_lds && cfg.hist_lds && (cfg.traits & kTrFlatGeneral) == kTrFlatGeneral)  // ... with triangle leaves among the top level's children
        k = cfg.per_cu <= 3 ? k_fused<true, true, false, 3, false, false, false, kTrFlatGeneral> : k_fused<true, true, false, MTR_FUSED_MIN_WAVES, false, false, false, kTrFlatGeneral>;
    else if (!NLOS && cfg.scene_lds && cfg.hist_lds && (cfg.traits & kTrCornell) == kTrCornell)          // diffuse materials, one rectangle emitter: the specialised shading code
        k = cfg.per_cu <= 3 ? k_fused<true, true, false, 3, false, false, false, kTrCornell> : k_fused<true, true, false, MTR_FUSED_MIN_WAVES, false, false, false, kTrCornell>;
    // (NOT for the NLOS loop, whose scene — a relay wall and three triangle pairs — is one flat node: its rays are coherent, camera ->
    // wall -> hidden geometry, and walk together; with the flat walk config 4's share took 10.5 instead of 8.8 ms, measured in round 6)
    else if (NLOS && (cfg.traits & kTrGrey)) {      // a grey NLOS scene: one plane per row (fused_plan)
        if constexpr (NLOS) k = k_fused<true, true, true, 3, false, false, false, kTrGrey>;
    }
    else if (cfg.scene_lds && cfg.hist_lds) {       // (the NLOS loop: at most three workgroups per CU — fused_plan — hence the 168-register form only)
        if constexpr (NLOS) k = k_fused<true, true, true, 3>;
        else k = cfg.per_cu <= 3 ? k_fused<true, true, false, 3> : k_fused<true, true, false>;
    }
    else if (cfg.scene_lds && !cfg.hist_lds) k = k_fused<true, false, NLOS>;
    else if (!cfg.scene_lds && cfg.hist_lds) k = k_fused<false, true, NLOS>;
    else k = k_fused<false, false, NLOS>;
#endif
    hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cfg.lds_bytes);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(args.ticket, 0, sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(cfg.grid), dim3(kBlock), cfg.lds_bytes, stream, args);
    return hipGetLastError();
}

// NLOS prepare: scanned points of every film pixel + the point the laser's axis hits (Single capture)
__global__ void __launch_bounds__(kBlock) k_nlos_prepare(SceneDev sc, NlosConst nc, q4 *targets)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    LdsStack st; st.base = (int32_t *)smem + threadIdx.x; st.sp = 0;
    SceneView sv;
    sv.nodes = sc.nodes; sv.tpairs = sc.tpairs; sv.tshade = sc.tshade; sv.mats = sc.mats; sv.ems = sc.ems;
    sv.wnodes = nullptr; sv.wnodes4 = nullptr; sv.wnodes8q = nullptr;
    sv.node_pairs = false;
    sv.n_emitters = sc.n_ems; sv.n_slots = sc.n_slots;
    sv.samp_tris = sc.samp_tris; sv.samp_vn = sc.samp_vn; sv.face_pmf = sc.face_pmf; sv.face_cdf = sc.face_cdf; sv.vnormals = sc.vnormals;
    sv.texels = sc.texels; sv.tex_info = sc.tex_info; sv.uvs = sc.uvs;
    const uint32_t total = nlos_target_count(nc);
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < total; i += gridDim.x * kBlock) {
        const Ray r = nlos_prepare_ray(nc, i);
        const Hit h = traverse<false>(sv, r.o, r.d, r.tmax, st);
        f3 p = mk(0, 0, 0);
        if (h.prim >= 0) p = hit_ctx<false>(sv, r.d, h).sp;
        targets[i] = q4{ p.x, p.y, p.z, h.prim >= 0 ? 1.0f : 0.0f };
    }
}

hipError_t launch_nlos_prepare(const SceneDev &sc, const NlosConst &nc, q4 *targets, hipStream_t stream)
{
    const size_t lds = (size_t)65 * kBlock * 4;
    hipError_t e = hipFuncSetAttribute((const void *)k_nlos_prepare, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    uint32_t n = nlos_target_count(nc);
    uint32_t blocks = (n + kBlock - 1) / kBlock; if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_nlos_prepare, dim3(blocks), dim3(kBlock), lds, stream, sc, nc, targets);
    return hipGetLastError();
}

hipError_t launch_fused(const FusedArgs &args, const FusedConfig &cfg, hipStream_t stream)
{
    return args.nlos_on ? launch_fused_s<true>(args, cfg, stream) : launch_fused_s<false>(args, cfg, stream);
}

// ------------------------------------------------------------------ stand-alone time-bin scatter-add
// variant 0: one f32 atomic per channel straight into HBM (the contract form of
// transient_image_block.py:148-149)
__global__ void __launch_bounds__(kBlock) k_splat_atomic(mtr_splat_soa s, Film film, float *out, DevCounters *cnt,
                                                         const uint32_t *only_if)
{
    if (only_if && *only_if == 0u) return;          // the fallback of variant 1: runs only when the input was not sorted
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    const uint32_t npix = film.width * film.height;
    uint32_t mine = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < s.n; i += stride) {
        const uint32_t pixel = s.pixel[i];
        const int32_t bin = film_row_bin(film, s.opl[i], s.laser ? s.laser[i] : 0u);
        if (bin < 0 || pixel >= npix) continue;
        const size_t idx = ((size_t)pixel * film.bins + (uint32_t)bin) * 4u;
        unsafeAtomicAdd(out + idx, s.r[i]); unsafeAtomicAdd(out + idx + 1, s.g[i]); unsafeAtomicAdd(out + idx + 2, s.b[i]);
        ++mine;
    }
    if (cnt && mine) atomicAdd(&cnt->splats_issued, (unsigned long long)mine);
}

// variant 1: contributions sorted by pixel.  k_splat_runs finds where each pixel's run starts (and whether the input is
// sorted at all); k_splat_rows then works like k_wf_scatter: one workgroup per pixel streams the run into an LDS row —
// 64-bit fixed point, ds_add_u64 (ds_add_f32 retires at 3 clocks per lane on gfx950) — and adds the touched bins to the
// film with plain 16-byte read-modify-writes, because the pixel is its own.  Unsorted input is partitioned by pixel first
// (mtr_splat.hip) or, where that cannot be done, falls back to the atomics.
//
// The run table is SPARSE: it is preset to kNoRun, and the record that opens a run (pixel[i - 1] < pixel[i]) writes two
// entries — the end of the previous pixel's run, starts[pixel[i - 1] + 1], and the start of its own, starts[pixel[i]]; the
// pixels in between have no contributions and stay kNoRun.  (Rounds 1-3 FILLED the gap: a few entries per record on sorted
// input — and a hundred thousand per record on unsorted input, where every second record opens a "gap": 52 s for 2^28
// uniform contributions, found when the partition path first ran this kernel on such input.)
constexpr unsigned long long kNoRun = ~0ull;
__global__ void __launch_bounds__(kBlock) k_splat_runs(const uint32_t *pixel, uint64_t n, uint32_t npix,
                                                       unsigned long long *starts, uint32_t *unsorted)
{
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    bool descent = false;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const uint32_t cur = min(pixel[i], npix);                    // ids >= npix (dropped contributions) sort to the end
        const int64_t prev = i ? (int64_t)min(pixel[i - 1], npix) : -1;
        descent = descent || (int64_t)cur < prev;
        if (!descent && (int64_t)cur > prev) { starts[prev + 1] = i; starts[cur] = i; }       // (after a descent the table is void: no more writes)
        if (i == n - 1 && cur < npix) starts[cur + 1] = n;
    }
    // one store per wave that saw a descent (a store per thread — half a million to one address on random input — is slow)
    if (__ballot(descent) != 0ull && (threadIdx.x & 63u) == 0u) *unsorted = 1u;
}

template <bool FIXED>
__global__ void __launch_bounds__(kBlock) k_splat_rows(mtr_splat_soa s, Film film, float *out, const unsigned long long *starts,
                                                       const uint32_t *unsorted, DevCounters *cnt, uint32_t film_zero)
{
    if (*unsorted) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ uint32_t s_redo;                                      // (FIXED) RANGE GUARD: this pixel's row must be rebuilt in f32
    float *row = (float *)smem;                                      // [3][T] f32 ...
    unsigned long long *row64 = (unsigned long long *)smem;          // ... or [3][T] 2^-42 fixed point
    const uint32_t T = film.bins, npix = film.width * film.height;
    const int tid = threadIdx.x;
    if (tid == 0) s_redo = 0u;
    for (uint32_t t = tid; t < 3 * T; t += kBlock) { if (FIXED) row64[t] = 0ull; else row[t] = 0.0f; }
    __syncthreads();
    uint32_t mine = 0;
    // as k_wf_scatter: a batch of independent loads in flight per lane (4 records x 4 arrays), and the first batch of the NEXT
    // pixel's run is requested before this pixel's row is flushed (contributions and film rows are touched once: non-temporal)
    constexpr int kBatch = 4;
    float bo[kBatch], br[kBatch], bg[kBatch], bb[kBatch]; uint32_t bl[kBatch]; bool bv[kBatch];
    auto fetch = [&](uint64_t lo_, uint64_t hi_) {
#pragma unroll
        for (int k = 0; k < kBatch; ++k) {
            const uint64_t i = lo_ + (uint64_t)k * kBlock + tid;
            bv[k] = i < hi_; bo[k] = 0.0f; br[k] = bg[k] = bb[k] = 0.0f; bl[k] = 0u;
            if (bv[k]) {
                bo[k] = __builtin_nontemporal_load(s.opl + i); br[k] = __builtin_nontemporal_load(s.r + i);
                bg[k] = __builtin_nontemporal_load(s.g + i); bb[k] = __builtin_nontemporal_load(s.b + i);
                if (s.laser) bl[k] = s.laser[i];
            }
        }
    };
    // a pixel WITH a run has both table entries set — its start by the record that opens it, its end by the opener of the next
    // run or by the last record; the pixel right behind a run has only its first entry set (that run's end), the rest of a gap
    // neither: those are empty
    auto run_of = [&](uint32_t px_, uint64_t &lo_, uint64_t &hi_) {
        lo_ = starts[px_]; hi_ = starts[px_ + 1];
        if (lo_ == kNoRun || hi_ == kNoRun) { lo_ = 0; hi_ = 0; }
    };
    uint64_t lo_n = 0, hi_n = 0;
    if (blockIdx.x < npix) { run_of(blockIdx.x, lo_n, hi_n); fetch(lo_n, hi_n); }
    for (uint32_t px = blockIdx.x; px < npix; px += gridDim.x) {
        const uint64_t lo = lo_n, hi = hi_n;
        const uint32_t px_next = px + gridDim.x;
        if (px_next < npix) run_of(px_next, lo_n, hi_n);
        const float lim = splat_fixed_limit(hi - lo);
        bool unsafe = false;
        for (uint64_t base = lo; base < hi || base == lo; base += (uint64_t)kBatch * kBlock) {
            if (base != lo) fetch(base, hi);
#pragma unroll
            for (int k = 0; k < kBatch; ++k) {
                if (!bv[k]) continue;
                const int32_t bin = film_row_bin(film, bo[k], bl[k]);
                if (bin < 0) continue;
                if (FIXED) {
                    unsigned long long *p = row64 + bin;
                    unsafe |= splat_fixed_unsafe(br[k], bg[k], bb[k], lim);
                    atomicAdd(p, splat_to_fixed(br[k])); atomicAdd(p + T, splat_to_fixed(bg[k])); atomicAdd(p + 2 * T, splat_to_fixed(bb[k]));
                } else {
                    lds_add(row + bin, br[k]); lds_add(row + T + bin, bg[k]); lds_add(row + 2 * T + bin, bb[k]);
                }
                ++mine;
            }
        }
        if (px_next < npix) fetch(lo_n, hi_n);            // in flight across the flush below
        if (lo == hi) continue;                           // no contributions (uniform across the workgroup)
        if (FIXED && unsafe) s_redo = 1u;
        __syncthreads();
        bool as_f32 = !FIXED;
        if (FIXED && s_redo != 0u) {
            // RANGE GUARD: a value of this run does not fit the fixed-point row — the row again, in f32 (the next run's first batch
            // stays in the batch registers: this block has its own loads; `mine` was counted by the first pass)
            as_f32 = true;
            for (uint32_t t = tid; t < 6 * T; t += kBlock) row[t] = 0.0f;
            __syncthreads();
            if (tid == 0) s_redo = 0u;
            for (uint64_t i = lo + tid; i < hi; i += kBlock) {
                const int32_t bin = film_row_bin(film, s.opl[i], s.laser ? s.laser[i] : 0u);
                if (bin < 0) continue;
                lds_add(row + bin, s.r[i]); lds_add(row + T + bin, s.g[i]); lds_add(row + 2 * T + bin, s.b[i]);
            }
            __syncthreads();
        }
        float4 *dst = (float4 *)(out + (size_t)px * T * 4u);
        for (uint32_t t = tid; t < T; t += kBlock) {
            float r, g, b; bool nz;
            if (FIXED && !as_f32) {
                const unsigned long long qr = row64[t], qg = row64[T + t], qb = row64[2 * T + t];
                nz = (qr | qg | qb) != 0ull;
                r = __ll2float_rn((long long)qr) * 2.2737367544323206e-13f; g = __ll2float_rn((long long)qg) * 2.2737367544323206e-13f;
                b = __ll2float_rn((long long)qb) * 2.2737367544323206e-13f;
                if (nz) { row64[t] = 0ull; row64[T + t] = 0ull; row64[2 * T + t] = 0ull; }
            } else {
                r = row[t]; g = row[T + t]; b = row[2 * T + t];
                nz = r != 0.0f || g != 0.0f || b != 0.0f;
                if (nz) { row[t] = 0.0f; row[T + t] = 0.0f; row[2 * T + t] = 0.0f; }
            }
            // film_zero (MTR_SPLAT_FILM_ZERO): the caller vouches that the film is zero — the row goes out whole, no read
            if (film_zero) nt_store(dst + t, make_float4(r, g, b, 0.0f));
            else if (nz) { float4 v = nt_load(dst + t); v.x += r; v.y += g; v.z += b; nt_store(dst + t, v); }
        }
        __syncthreads();
    }
    if (cnt && mine) atomicAdd(&cnt->splats_issued, (unsigned long long)mine);
}

__global__ void k_splat_phasor(mtr_splat_soa s, Film film, float *out, DevCounters *cnt);

// fallback_atomics: variant 1 on unsorted input falls back, on the device, to the f32 atomics (false: the caller partitions the
// input by pixel instead — launch_splat_partitioned — after reading the flag at scratch[0])
hipError_t launch_splat_add(int variant, const mtr_splat_soa &s, const Film &film, float *film_out,
                            DevCounters *counters, void *scratch, hipStream_t stream, bool film_zero, bool fallback_atomics)
{
    if (s.n == 0) return hipSuccess;
    if (film.n_freq) {
        uint64_t blocks = (s.n + kBlock - 1) / kBlock;
        if (blocks > 256 * 16) blocks = 256 * 16;
        hipLaunchKernelGGL(k_splat_phasor, dim3((unsigned)blocks), dim3(kBlock), 0, stream, s, film, film_out, counters);
        return hipGetLastError();
    }
    uint64_t blocks = (s.n + kBlock - 1) / kBlock;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (variant == 0 || (size_t)film.bins * 12u > 150u * 1024u || !scratch) {
        hipLaunchKernelGGL(k_splat_atomic, dim3((unsigned)blocks), dim3(kBlock), 0, stream, s, film, film_out, counters, nullptr);
    } else {
        const uint32_t npix = film.width * film.height;
        uint32_t *unsorted = (uint32_t *)scratch;                                    // [0]: flag; [2..]: run starts (u64[npix + 1])
        unsigned long long *starts = (unsigned long long *)scratch + 1;
        hipError_t e = hipMemsetAsync(unsorted, 0, 8, stream);
        if (e != hipSuccess) return e;
        e = hipMemsetAsync(starts, 0xff, 8u * ((size_t)npix + 1u), stream);          // kNoRun
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_splat_runs, dim3((unsigned)blocks), dim3(kBlock), 0, stream, s.pixel, (uint64_t)s.n, npix, starts, unsorted);
        const bool fixed = (size_t)film.bins * 24u <= 72u * 1024u;
        const size_t lds = (size_t)film.bins * (fixed ? 24u : 12u);
        int per_cu = (int)((150u * 1024u) / (lds + 64)); if (per_cu > 8) per_cu = 8; if (per_cu < 1) per_cu = 1;
        const unsigned grid = npix < (uint32_t)(256 * per_cu) ? npix : (unsigned)(256 * per_cu);
        void (*k)(mtr_splat_soa, Film, float *, const unsigned long long *, const uint32_t *, DevCounters *, uint32_t) =
            fixed ? k_splat_rows<true> : k_splat_rows<false>;
        e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3(grid), dim3(kBlock), lds, stream, s, film, film_out, starts, unsorted, counters, film_zero ? 1u : 0u);
        if (fallback_atomics)
            hipLaunchKernelGGL(k_splat_atomic, dim3((unsigned)blocks), dim3(kBlock), 0, stream, s, film, film_out, counters, unsorted);
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------ develop (transient_hdr_film.py:220-248)
__global__ void __launch_bounds__(kBlock) k_develop_transient(const float4 *__restrict__ in, float *__restrict__ out, uint64_t n)
{
    // each thread converts 4 consecutive (R,G,B,W) texels into 12 contiguous floats: 3 x 16-byte stores
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    const uint64_t n4 = n / 4;
    for (uint64_t q = (uint64_t)blockIdx.x * kBlock + threadIdx.x; q < n4; q += stride) {
        float4 a = in[4 * q], b = in[4 * q + 1], c = in[4 * q + 2], d = in[4 * q + 3];
        float wa = a.w == 0.0f ? 1.0f : a.w, wb = b.w == 0.0f ? 1.0f : b.w;
        float wc = c.w == 0.0f ? 1.0f : c.w, wd = d.w == 0.0f ? 1.0f : d.w;
        float4 o0 = make_float4(a.x / wa, a.y / wa, a.z / wa, b.x / wb);
        float4 o1 = make_float4(b.y / wb, b.z / wb, c.x / wc, c.y / wc);
        float4 o2 = make_float4(c.z / wc, d.x / wd, d.y / wd, d.z / wd);
        float4 *o = (float4 *)(out + 12 * q);
        o[0] = o0; o[1] = o1; o[2] = o2;
    }
    // tail (n not a multiple of 4)
    for (uint64_t i = n4 * 4 + (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        float4 a = in[i];
        float w = a.w == 0.0f ? 1.0f : a.w;
        out[3 * i] = a.x / w; out[3 * i + 1] = a.y / w; out[3 * i + 2] = a.z / w;
    }
}
__global__ void __launch_bounds__(kBlock) k_develop_steady(const float4 *__restrict__ in, float *__restrict__ out, uint64_t n)
{
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        float4 a = in[i];
        bool ok = a.w != 0.0f;
        out[3 * i] = ok ? a.x / a.w : 0.0f; out[3 * i + 1] = ok ? a.y / a.w : 0.0f; out[3 * i + 2] = ok ? a.z / a.w : 0.0f;
    }
}

// phasor film from Python (add_transient_data): one thread per contribution, 2F atomics
__global__ void __launch_bounds__(kBlock) k_splat_phasor(mtr_splat_soa s, Film film, float *out, DevCounters *cnt)
{
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    const uint32_t npix = film.width * film.height, F = film.n_freq;
    uint32_t mine = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < s.n; i += stride) {
        const uint32_t pixel = s.pixel[i];
        const float opl = s.opl[i];
        if (pixel >= npix || film_bin(film, opl) < 0) continue;
        const float rel = opl - film.start_opl, v = s.r[i];
        float *dst = out + (size_t)pixel * (2u * F + 1u);
        for (uint32_t f = 0; f < F; ++f) {
            float c, sn;
            phasor_term(film.freq[f], rel, c, sn);
            unsafeAtomicAdd(dst + 2 * f, v * c); unsafeAtomicAdd(dst + 2 * f + 1, v * sn);
        }
        ++mine;
    }
    if (cnt && mine) atomicAdd(&cnt->splats_issued, (unsigned long long)mine);
}

// develop_phasors_ (phasor_hdr_film.py:216-238): (H,W,2F+1) -> (H,W,F,2), values / (weight == 0 ? 1 : weight)
__global__ void __launch_bounds__(kBlock) k_develop_phasor(const float *in, float *out, uint64_t n_pix, uint32_t F)
{
    const uint64_t n = n_pix * 2u * F, stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const uint64_t px = i / (2u * F);
        const float w = in[px * (2u * F + 1u) + 2u * F];
        out[i] = in[px * (2u * F + 1u) + (i - px * 2u * F)] / (w == 0.0f ? 1.0f : w);
    }
}

hipError_t launch_develop(const Film &film, const float *t4, float *t3, const float *s4, float *s3, hipStream_t stream)
{
    if (t4 && t3 && film.n_freq) {
        const uint64_t npx = (uint64_t)film.width * film.height;
        uint64_t blocks = (npx * 2u * film.n_freq + kBlock - 1) / kBlock; if (blocks > 256 * 8) blocks = 256 * 8; if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(k_develop_phasor, dim3((unsigned)blocks), dim3(kBlock), 0, stream, t4, t3, npx, film.n_freq);
    } else if (t4 && t3) {
        uint64_t n = (uint64_t)film.width * film.height * film.bins;
        uint64_t blocks = (n / 4 + kBlock - 1) / kBlock; if (blocks > 256 * 8) blocks = 256 * 8; if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(k_develop_transient, dim3((unsigned)blocks), dim3(kBlock), 0, stream, (const float4 *)t4, t3, n);
    }
    if (s4 && s3) {
        uint64_t n = (uint64_t)film.width * film.height;
        uint64_t blocks = (n + kBlock - 1) / kBlock; if (blocks > 256 * 8) blocks = 256 * 8; if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(k_develop_steady, dim3((unsigned)blocks), dim3(kBlock), 0, stream, (const float4 *)s4, s3, n);
    }
    return hipGetLastError();
}

} // namespace mtr
