// mtr_bvh.cpp — binned-SAH BVH2 over triangles, flattened to "node packets": one record per
// inner node holding BOTH children's (padded) boxes and references, so a traversal step is one
// 64-byte fetch (4 x ds_read_b128 from LDS, or one cache line from L2) and two slab tests.
//
// Build primitives ("items"): a mesh triangle; an analytic RECTANGLE (its two carrier triangles, always alone in its
// leaf); an OBJECT — a small mesh shape with a known object -> world transform, kept together in one subtree whose
// splits are chosen in the shape's OBJECT space (where the faces of a rotated cube are flat: coplanar triangles end up in
// the same leaf) and which the 8-wide collapse turns into one object node with object-space boxes (mtr_core.h, WNodeT).
#include "mtr_bvh.h"
#include <thread>
#include <chrono>
#include <atomic>
#include "mtr_knobs.h"

#include <algorithm>
#include <cmath>
#include <cfloat>
#include <cstdio>
#include <cstdlib>

namespace mtr {
namespace {

struct Box {
    float lo[3], hi[3];
    void reset() { for (int k = 0; k < 3; ++k) { lo[k] = FLT_MAX; hi[k] = -FLT_MAX; } }
    void grow(const float *p) { for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], p[k]); hi[k] = std::max(hi[k], p[k]); } }
    void grow(const Box &b) { for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], b.lo[k]); hi[k] = std::max(hi[k], b.hi[k]); } }
    float area() const
    {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (dx < 0 || dy < 0 || dz < 0) return 0.0f;
        return 2.0f * (dx * dy + dy * dz + dz * dx);
    }
};

enum : uint8_t { kItemTri = 0, kItemQuad = 1, kItemObject = 2 };
struct Item { uint32_t first_tri, n_tris; uint8_t type; int32_t object; };

// world box of a triangle / its image under a 3 x 4 affine map (f64, then outward-rounded to f32 by the caller's padding)
Box tri_box(const float *v, const float *xf)
{
    Box b; b.reset();
    for (int k = 0; k < 3; ++k) {
        const float *p = v + 3 * k;
        if (!xf) { b.grow(p); continue; }
        float q[3];
        for (int r = 0; r < 3; ++r)
            q[r] = (float)((double)xf[4 * r] * p[0] + (double)xf[4 * r + 1] * p[1] + (double)xf[4 * r + 2] * p[2] + (double)xf[4 * r + 3]);
        b.grow(q);
    }
    return b;
}

// box of the part of polygon `p` (n vertices, f64) on one side of the plane x[axis] = pos  (Sutherland-Hodgman)
int clip_poly(const double (*p)[3], int n, int axis, double pos, bool keep_low, double (*out)[3])
{
    int m = 0;
    for (int i = 0; i < n; ++i) {
        const double *a = p[i], *b = p[(i + 1) % n];
        const bool ia = keep_low ? a[axis] <= pos : a[axis] >= pos, ib = keep_low ? b[axis] <= pos : b[axis] >= pos;
        if (ia) { for (int k = 0; k < 3; ++k) out[m][k] = a[k]; ++m; }
        if (ia != ib) {
            const double t = (pos - a[axis]) / (b[axis] - a[axis]);
            for (int k = 0; k < 3; ++k) out[m][k] = a[k] + t * (b[k] - a[k]);
            out[m][axis] = pos; ++m;
        }
    }
    return m;
}

// f32 box of a polygon of f64 vertices; the conversion rounds to nearest: widened by one part in 10^6 so that the piece stays inside
Box poly_box(const double (*p)[3], int n)
{
    Box b; b.reset();
    for (int k = 0; k < n; ++k) { const float q[3] = { (float)p[k][0], (float)p[k][1], (float)p[k][2] }; b.grow(q); }
    for (int k = 0; k < 3; ++k) { const float pad = 1e-6f * (1.0f + std::max(std::fabs(b.lo[k]), std::fabs(b.hi[k]))); b.lo[k] -= pad; b.hi[k] += pad; }
    return b;
}
Box intersect(const Box &a, const Box &b)
{
    Box r;
    for (int k = 0; k < 3; ++k) { r.lo[k] = std::max(a.lo[k], b.lo[k]); r.hi[k] = std::min(a.hi[k], b.hi[k]); }
    return r;
}
// triangle `v` (9 floats) clipped to box `b`: a convex polygon of at most 9 vertices (0: nothing left)
int tri_in_box(const float *v, const Box &b, double (*out)[3])
{
    double a[16][3], c[16][3];
    int n = 3;
    for (int k = 0; k < 3; ++k) for (int r = 0; r < 3; ++r) a[k][r] = v[3 * k + r];
    for (int ax = 0; ax < 3 && n >= 3; ++ax) {
        n = clip_poly(a, n, ax, (double)b.lo[ax], false, c);
        if (n < 3) break;
        n = clip_poly(c, n, ax, (double)b.hi[ax], true, a);
    }
    if (n < 3) return 0;
    for (int k = 0; k < n; ++k) for (int r = 0; r < 3; ++r) out[k][r] = a[k][r];
    return n;
}

struct Tmp { Box box; int left = -1, right = -1; uint32_t first = 0, count = 0; bool quad = false; int32_t object = -1; };

struct Shared {                       // what every (sub-)builder appends to
    const float *verts = nullptr;
    std::vector<Tmp> tmp;
    std::vector<uint32_t> leaf_tris;  // Tmp leaves: triangles [first, first + count) of this array
};

struct Builder {
    Shared &S;
    std::vector<Item> items;
    std::vector<Box> sbox;            // boxes the SPLITS are chosen on (object space inside an object, world otherwise)
    std::vector<Box> wbox;            // world boxes (what the nodes store)
    std::vector<float> cent;          // 3 per item, of sbox
    std::vector<uint32_t> order;
    const BvhPrims *prims = nullptr;

    static constexpr int kMaxBins = 64;
    int kBins = 16;
    uint32_t kLeafTarget = 2, kLeafMax = 4;
    // SPATIAL SPLITS (Stich, Friedrich, Dietrich 2009): where the two children of the best object split overlap, a node may instead be
    // cut by a PLANE; a triangle reference that straddles it is duplicated, each copy with the box of its part.  World-space builder over
    // large scenes only (build_bvh); `dup_budget` bounds the duplicates of the whole build.
    bool spatial = false;
    long long *dup_budget = nullptr;  // duplicates this builder may still make (the root builder draws on the build's budget in build order;
                                      // what is left when the subtrees are deferred is shared out among them by their reference counts —
                                      // build_parallel_finish — so that the tree does not depend on the workers' timing; the staircase
                                      // uses 0.56 n of its n)
    float root_area = 0.0f;
    static constexpr int kMaxSpatialBins = 64;
    int kSpatialBins = 16;
    // ... only where the object split's children overlap by more than this share of the scene's area (the paper's alpha; its 1e-5
    // duplicates 12 % of the staircase's triangles, 1e-6 52 %: config 5 at 256 spp k_wf_trace 117.2 / 113.5 ms, 3e-7 113.0, 0 (budget 2 n) 120)
    float kAlpha = 1e-6f;
    uint32_t kDepthBudget = 61;       // (build(): DEPTH BUDGET; MTR_BVH_DEPTH_BUDGET in test / experiment builds)
    bool kUnsplit = false;            // reference unsplitting (below): measured, k_wf_trace 112.1 with against 110.5 ms without — off

    explicit Builder(Shared &s) : S(s) {}

    void add_item(const Item &it, const float *xf)
    {
        Box sb, wb; sb.reset(); wb.reset();
        for (uint32_t t = 0; t < it.n_tris; ++t) {
            const float *v = S.verts + 9 * (size_t)(it.first_tri + t);
            wb.grow(tri_box(v, nullptr));
            sb.grow(tri_box(v, xf));
        }
        items.push_back(it); sbox.push_back(sb); wbox.push_back(wb);
        for (int k = 0; k < 3; ++k) cent.push_back(0.5f * (sb.lo[k] + sb.hi[k]));
        order.push_back((uint32_t)items.size() - 1u);
    }

    // a REFERENCE to a mesh triangle with its own box: a piece of a large triangle (early split clipping / spatial splits, below)
    uint32_t add_reference(uint32_t tri, const Box &b)
    {
        items.push_back(Item{ tri, 1u, kItemTri, -1 }); sbox.push_back(b); wbox.push_back(b);
        for (int k = 0; k < 3; ++k) cent.push_back(0.5f * (b.lo[k] + b.hi[k]));
        order.push_back((uint32_t)items.size() - 1u);
        return (uint32_t)items.size() - 1u;
    }
    void set_box(uint32_t t, const Box &b)
    {
        sbox[t] = b; wbox[t] = b;
        for (int k = 0; k < 3; ++k) cent[3 * (size_t)t + k] = 0.5f * (b.lo[k] + b.hi[k]);
    }

    int make_leaf(const std::vector<uint32_t> &refs)
    {
        Tmp t; t.box.reset(); t.first = (uint32_t)S.leaf_tris.size();
        for (uint32_t r : refs) {
            const Item &it = items[r];
            t.box.grow(wbox[r]);
            t.quad = it.type == kItemQuad;
            for (uint32_t k = 0; k < it.n_tris; ++k) {
                bool dup = false;                         // two pieces of one triangle in the same leaf: one test
                for (size_t j = t.first; j < S.leaf_tris.size(); ++j) dup = dup || S.leaf_tris[j] == it.first_tri + k;
                if (!dup) S.leaf_tris.push_back(it.first_tri + k);
            }
        }
        t.count = (uint32_t)S.leaf_tris.size() - t.first;
        S.tmp.push_back(t);
        return (int)S.tmp.size() - 1;
    }

    // an object: its own sub-tree, splits chosen on object-space boxes
    int build_object(const Item &it)
    {
        const float *xf = prims->object_xf + 12 * (size_t)it.object;
        Builder B(S);
        B.kLeafTarget = kLeafTarget; B.kLeafMax = 2;           // object-space leaves: at most one coplanar pair
        B.kDepthBudget = kDepthBudget;
        for (uint32_t t = 0; t < it.n_tris; ++t) B.add_item(Item{ it.first_tri + t, 1u, kItemTri, -1 }, xf);
        const int root = B.build(B.order);
        S.tmp[root].object = it.object;
        return root;
    }

    // what n triangles cost a ray that enters their box.  The large scenes' leaves are tested a PAIR at a time (packed Moeller-Trumbore):
    // counting pairs makes the SAH prefer 2 + 2 to 3 + 1 at the bottom of the tree (config 5 at 256 spp, k_wf_trace 110.4 -> 108.5 ms;
    // the exact sweep over every split of nodes of <= 16 / 64 references instead of bins changed nothing on top: 108.6 / 108.4 ms)
    bool kPairCost = false;
    float tests(uint32_t n) const { return kPairCost ? (float)((n + 1u) / 2u) : (float)n; }

    bool splittable(uint32_t t) const { return items[t].type == kItemTri && items[t].n_tris == 1u; }

    // the parts of reference t on either side of the plane x[axis] = pos (boxes inside the reference's own box); false: the plane
    // leaves nothing on one side
    bool split_reference(uint32_t t, int axis, float pos, Box &lo_b, Box &hi_b) const
    {
        double poly[16][3], lo_p[16][3], hi_p[16][3];
        const int n = tri_in_box(S.verts + 9 * (size_t)items[t].first_tri, sbox[t], poly);
        if (n < 3) return false;
        const int nl = clip_poly(poly, n, axis, (double)pos, true, lo_p), nh = clip_poly(poly, n, axis, (double)pos, false, hi_p);
        if (nl < 3 || nh < 3) return false;
        lo_b = intersect(poly_box(lo_p, nl), sbox[t]); hi_b = intersect(poly_box(hi_p, nh), sbox[t]);
        lo_b.hi[axis] = std::min(lo_b.hi[axis], std::nextafter(pos, INFINITY)); hi_b.lo[axis] = std::max(hi_b.lo[axis], std::nextafter(pos, -INFINITY));
        for (int k = 0; k < 3; ++k) if (lo_b.lo[k] > lo_b.hi[k] || hi_b.lo[k] > hi_b.hi[k]) return false;
        return true;
    }

    // PARALLEL BUILD (round 6).  The root builder splits nodes until a subtree holds at most `defer_grain` references; such a subtree is
    // DEFERRED — a placeholder node now, built later by a worker thread in a builder of its own (copies of its references, its own
    // node and leaf arrays) and stitched in behind the placeholder.  Every split is a function of the node's references alone, so the
    // tree is the one the sequential build makes, node for node (tests/test_scene_host.py), as long as the duplicate budget lasts.
    struct Deferred { int tmp; std::vector<uint32_t> refs; uint32_t depth; };
    std::vector<Deferred> deferred;
    std::vector<int> top_inner;       // inner nodes made by the root builder, children first: their boxes are set once the placeholders are real
    size_t defer_grain = 0;
    unsigned n_threads = 1;

    void build_parallel_finish()
    {
        std::vector<Shared> LS(deferred.size());
        std::vector<int> roots(deferred.size(), -1);
        // every subtree's share of the duplicates that are left
        std::vector<long long> share(deferred.size(), 0);
        if (dup_budget) {
            size_t total = 0;
            for (const Deferred &D : deferred) total += D.refs.size();
            // (four times the even share: where the long thin triangles are, a subtree needs more duplicates than it has references — with
            // even shares the staircase made 91 k duplicates instead of the 146 k of one budget drawn on in build order; the bound on
            // memory becomes 4 n in the worst case, 0.56 n in this scene)
            for (size_t i = 0; i < deferred.size(); ++i) share[i] = total ? (long long)(4.0 * (double)*dup_budget * (double)deferred[i].refs.size() / (double)total) : 0;
        }
        // largest subtrees first (they finish last otherwise)
        std::vector<size_t> by_size(deferred.size());
        for (size_t i = 0; i < by_size.size(); ++i) by_size[i] = i;
        std::stable_sort(by_size.begin(), by_size.end(), [&](size_t a, size_t b) { return deferred[a].refs.size() > deferred[b].refs.size(); });
        std::atomic<size_t> next{ 0 };
        auto work = [&]() {
            for (;;) {
                const size_t k = next.fetch_add(1);
                if (k >= by_size.size()) break;
                const size_t i = by_size[k];
                const Deferred &D = deferred[i];
                LS[i].verts = S.verts;
                Builder LB(LS[i]);
                LB.prims = prims; LB.kBins = kBins; LB.kLeafTarget = kLeafTarget; LB.kLeafMax = kLeafMax; LB.spatial = spatial;
                LB.dup_budget = dup_budget ? &share[i] : nullptr; LB.root_area = root_area; LB.kSpatialBins = kSpatialBins; LB.kAlpha = kAlpha; LB.kUnsplit = kUnsplit;
                LB.kPairCost = kPairCost; LB.kDepthBudget = kDepthBudget;
                const size_t m = D.refs.size();
                LB.items.reserve(2 * m); LB.sbox.reserve(2 * m); LB.wbox.reserve(2 * m); LB.cent.reserve(6 * m); LB.order.reserve(2 * m);
                for (uint32_t r : D.refs) {
                    LB.items.push_back(items[r]); LB.sbox.push_back(sbox[r]); LB.wbox.push_back(wbox[r]);
                    for (int c = 0; c < 3; ++c) LB.cent.push_back(cent[3 * (size_t)r + c]);
                    LB.order.push_back((uint32_t)LB.items.size() - 1u);
                }
                roots[i] = LB.build(LB.order, D.depth);
            }
        };
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < n_threads; ++t) pool.emplace_back(work);
        work();
        for (std::thread &t : pool) t.join();
        // stitch, in the order the subtrees were deferred
        for (size_t i = 0; i < deferred.size(); ++i) {
            const int base = (int)S.tmp.size();
            const uint32_t leaf_base = (uint32_t)S.leaf_tris.size();
            for (Tmp t : LS[i].tmp) {
                if (t.left >= 0) { t.left += base; t.right += base; } else t.first += leaf_base;
                S.tmp.push_back(t);
            }
            S.leaf_tris.insert(S.leaf_tris.end(), LS[i].leaf_tris.begin(), LS[i].leaf_tris.end());
            S.tmp[deferred[i].tmp] = S.tmp[base + roots[i]];
            LS[i] = Shared();
        }
        if (dup_budget) {          // (for the build's statistics: the budget minus what the subtrees used)
            size_t total = 0;
            for (const Deferred &D : deferred) total += D.refs.size();
            long long used = 0;
            for (size_t i = 0; i < deferred.size(); ++i) used += (long long)(4.0 * (double)*dup_budget * (double)deferred[i].refs.size() / (double)total) - share[i];
            *dup_budget -= used;
        }
        for (int k : top_inner) { Tmp &t = S.tmp[k]; t.box = S.tmp[t.left].box; t.box.grow(S.tmp[t.right].box); }
        deferred.clear(); top_inner.clear();
    }

    // the loops over the references of a LARGE node (the root builder's: everything above the deferred subtrees) run in chunks on
    // worker threads; f(chunk, begin, end); what the chunks produce — boxes, counts — is merged by min / max / integer sums, which
    // do not care about the order
    unsigned chunks_for(size_t n) const { return (n_threads > 1 && n >= 16384) ? n_threads : 1u; }
    template <class F> void par_chunks(size_t n, unsigned nc, F f) const
    {
        if (nc <= 1u) { f(0u, (size_t)0, n); return; }
        std::vector<std::thread> pool;
        for (unsigned c = 1; c < nc; ++c) pool.emplace_back([&, c] { f(c, n * c / nc, n * (c + 1) / nc); });
        f(0u, (size_t)0, n / nc);
        for (std::thread &t : pool) t.join();
    }

    int build(std::vector<uint32_t> refs, uint32_t depth = 0)
    {
        if (defer_grain && refs.size() <= defer_grain) {
            S.tmp.emplace_back();
            deferred.push_back(Deferred{ (int)S.tmp.size() - 1, std::move(refs), depth });
            return (int)S.tmp.size() - 1;
        }
        const uint32_t count = (uint32_t)refs.size();
        bool special = false;
        uint32_t n_tris = 0;
        for (uint32_t t : refs) { special |= items[t].type != kItemTri; n_tris += items[t].n_tris; }
        if (count == 1 && items[refs[0]].type == kItemObject) return build_object(items[refs[0]]);
        if (count == 1 || (!special && count <= kLeafTarget)) return make_leaf(refs);

        Box b, cb; b.reset(); cb.reset();
        for (uint32_t t : refs) { b.grow(sbox[t]); cb.grow(&cent[3 * (size_t)t]); }
        // binned SAH over the three axes
        float best_cost = FLT_MAX; int best_axis = -1, best_split = -1;
        for (int ax = 0; ax < 3; ++ax) {
            float ext = cb.hi[ax] - cb.lo[ax];
            if (!(ext > 0.0f)) continue;
            Box bins[kMaxBins]; uint32_t cnt[kMaxBins];
            for (int k = 0; k < kBins; ++k) { bins[k].reset(); cnt[k] = 0; }
            float scale = (float)kBins / ext;
            const unsigned nc = chunks_for(refs.size());
            struct ObjBins { Box bins[kMaxBins]; uint32_t cnt[kMaxBins]; };
            std::vector<ObjBins> part(nc > 1u ? nc : 0u);
            par_chunks(refs.size(), nc, [&](unsigned c, size_t lo_i, size_t hi_i) {
                Box *pb = nc > 1u ? part[c].bins : bins; uint32_t *pc = nc > 1u ? part[c].cnt : cnt;
                if (nc > 1u) for (int k = 0; k < kBins; ++k) { pb[k].reset(); pc[k] = 0; }
                for (size_t i = lo_i; i < hi_i; ++i) {
                    const uint32_t t = refs[i];
                    int k = std::min(kBins - 1, std::max(0, (int)((cent[3 * (size_t)t + ax] - cb.lo[ax]) * scale)));
                    pb[k].grow(sbox[t]); pc[k] += items[t].n_tris;
                }
            });
            for (const ObjBins &P : part) for (int k = 0; k < kBins; ++k) { bins[k].grow(P.bins[k]); cnt[k] += P.cnt[k]; }
            float right_area[kMaxBins]; uint32_t right_cnt[kMaxBins];
            Box acc; acc.reset(); uint32_t c = 0;
            for (int k = kBins - 1; k > 0; --k) { acc.grow(bins[k]); c += cnt[k]; right_area[k] = acc.area(); right_cnt[k] = c; }
            acc.reset(); c = 0;
            for (int k = 0; k < kBins - 1; ++k) {
                acc.grow(bins[k]); c += cnt[k];
                if (c == 0 || right_cnt[k + 1] == 0) continue;
                float cost = acc.area() * tests(c) + right_area[k + 1] * tests(right_cnt[k + 1]);
                if (cost < best_cost) { best_cost = cost; best_axis = ax; best_split = k; }
            }
        }
        auto object_side = [&](uint32_t t) {          // true: left child of the best object split
            const float ext = cb.hi[best_axis] - cb.lo[best_axis];
            const float scale = (float)kBins / ext;
            const int k = std::min(kBins - 1, std::max(0, (int)((cent[3 * (size_t)t + best_axis] - cb.lo[best_axis]) * scale)));
            return k <= best_split;
        };
        const float leaf_cost = b.area() * tests(n_tris);
        if (best_axis >= 0 && !special && n_tris <= kLeafMax && best_cost >= leaf_cost) return make_leaf(refs);   // a leaf is cheaper

        // ---- spatial split candidate: only where the children of the object split overlap noticeably
        int sp_axis = -1; float sp_pos = 0.0f, sp_cost = FLT_MAX;
        // (not below level 40: the walkers' stacks end at 64 levels, and a chain of planes that each shave a sliver off the same
        // references must not be what uses them up)
        if (spatial && dup_budget && *dup_budget > 0 && best_axis >= 0 && count > kLeafTarget && depth < 40u) {
            Box lb, rb; lb.reset(); rb.reset();
            for (uint32_t t : refs) (object_side(t) ? lb : rb).grow(sbox[t]);
            const Box ov = intersect(lb, rb);
            if (ov.area() > kAlpha * root_area) {
                for (int ax = 0; ax < 3; ++ax) {
                    const float lo = b.lo[ax], ext = b.hi[ax] - b.lo[ax];
                    if (!(ext > 0.0f)) continue;
                    Box bins[kMaxSpatialBins]; uint32_t n_in[kMaxSpatialBins], n_out[kMaxSpatialBins];
                    for (int k = 0; k < kSpatialBins; ++k) { bins[k].reset(); n_in[k] = 0; n_out[k] = 0; }
                    const float scale = (float)kSpatialBins / ext;
                    auto bin_of = [&](float x) { return std::min(kSpatialBins - 1, std::max(0, (int)((x - lo) * scale))); };
                    const unsigned nc = chunks_for(refs.size());
                    struct SpBins { Box bins[kMaxSpatialBins]; uint32_t n_in[kMaxSpatialBins], n_out[kMaxSpatialBins]; };
                    std::vector<SpBins> part(nc > 1u ? nc : 0u);
                    par_chunks(refs.size(), nc, [&](unsigned c, size_t lo_i, size_t hi_i) {
                        Box *pb = nc > 1u ? part[c].bins : bins; uint32_t *pi = nc > 1u ? part[c].n_in : n_in, *po = nc > 1u ? part[c].n_out : n_out;
                        if (nc > 1u) for (int k = 0; k < kSpatialBins; ++k) { pb[k].reset(); pi[k] = 0; po[k] = 0; }
                        for (size_t i = lo_i; i < hi_i; ++i) {
                            const uint32_t t = refs[i];
                            const Box &rb_ = sbox[t];
                            int k0 = bin_of(rb_.lo[ax]), k1 = bin_of(rb_.hi[ax]);
                            if (!splittable(t)) { k0 = k1 = bin_of(cent[3 * (size_t)t + ax]); }
                            pi[k0] += items[t].n_tris; po[k1] += items[t].n_tris;
                            if (k0 == k1) { pb[k0].grow(rb_); continue; }
                            double poly[16][3], piece[16][3], rest[16][3];
                            int n = tri_in_box(S.verts + 9 * (size_t)items[t].first_tri, rb_, poly);
                            if (n < 3) { for (int k = k0; k <= k1; ++k) pb[k].grow(rb_); continue; }
                            for (int k = k0; k <= k1 && n >= 3; ++k) {          // chop the polygon bin by bin
                                if (k == k1) { pb[k].grow(intersect(poly_box(poly, n), rb_)); break; }
                                const double edge = (double)lo + (double)(k + 1) * (double)ext / (double)kSpatialBins;
                                const int np_ = clip_poly(poly, n, ax, edge, true, piece);
                                if (np_ >= 3) pb[k].grow(intersect(poly_box(piece, np_), rb_));
                                n = clip_poly(poly, n, ax, edge, false, rest);
                                for (int q = 0; q < n; ++q) for (int r = 0; r < 3; ++r) poly[q][r] = rest[q][r];
                            }
                        }
                    });
                    for (const SpBins &P : part) for (int k = 0; k < kSpatialBins; ++k) { bins[k].grow(P.bins[k]); n_in[k] += P.n_in[k]; n_out[k] += P.n_out[k]; }
                    float right_area[kMaxSpatialBins]; uint32_t right_cnt[kMaxSpatialBins];
                    Box acc; acc.reset(); uint32_t c = 0;
                    for (int k = kSpatialBins - 1; k > 0; --k) { acc.grow(bins[k]); c += n_out[k]; right_area[k] = acc.area(); right_cnt[k] = c; }
                    acc.reset(); c = 0;
                    for (int k = 0; k < kSpatialBins - 1; ++k) {
                        acc.grow(bins[k]); c += n_in[k];
                        if (c == 0 || right_cnt[k + 1] == 0) continue;
                        const float cost = acc.area() * tests(c) + right_area[k + 1] * tests(right_cnt[k + 1]);
                        if (cost < sp_cost) { sp_cost = cost; sp_axis = ax; sp_pos = (float)((double)lo + (double)(k + 1) * (double)ext / (double)kSpatialBins); }
                    }
                }
            }
        }

        std::vector<uint32_t> left, right;
        // DEPTH BUDGET: the walkers' stacks end at 64 levels and a scene whose tree is deeper has no kernel at all.  When what
        // is left of the budget only just holds a BALANCED tree over this node's references, the node is split at the median of
        // the centroids on its longest axis, whatever the SAH says: depth + ceil(log2 count) stays below 62.
        uint32_t lg = 0; while ((1u << lg) < count) ++lg;
        if (depth + lg >= kDepthBudget && !special) {
            int ax = 0;
            for (int k = 1; k < 3; ++k) if (cb.hi[k] - cb.lo[k] > cb.hi[ax] - cb.lo[ax]) ax = k;
            std::vector<uint32_t> sorted(refs);
            std::stable_sort(sorted.begin(), sorted.end(), [&](uint32_t a, uint32_t b_) { return cent[3 * (size_t)a + ax] < cent[3 * (size_t)b_ + ax]; });
            left.assign(sorted.begin(), sorted.begin() + count / 2);
            right.assign(sorted.begin() + count / 2, sorted.end());
        } else
        if (sp_axis >= 0 && sp_cost < best_cost) {
            struct Cut { uint32_t t; Box lo_b, hi_b; };
            std::vector<Cut> cuts;
            long long left_budget = *dup_budget;
            for (uint32_t t : refs) {
                const Box &rb_ = sbox[t];
                if (rb_.hi[sp_axis] <= sp_pos) { left.push_back(t); continue; }
                if (rb_.lo[sp_axis] >= sp_pos) { right.push_back(t); continue; }
                Cut c; c.t = t;
                if (splittable(t) && left_budget > 0 && split_reference(t, sp_axis, sp_pos, c.lo_b, c.hi_b)) { cuts.push_back(c); --left_budget; }
                else (cent[3 * (size_t)t + sp_axis] < sp_pos ? left : right).push_back(t);
            }
            // (nothing has been changed yet: a plane that leaves one side empty falls back to the object split)
            if (left.size() + cuts.size() == 0u || right.size() + cuts.size() == 0u) { left.clear(); right.clear(); }
            else {
                // REFERENCE UNSPLITTING (the paper's 4.4): a straddling reference is cut only if that is cheaper than handing it whole
                // to one child — C_split = A(B1) N1 + A(B2) N2 against A(B1 + ref) N1 + A(B2) (N2 - 1) and its mirror image
                Box B1, B2; B1.reset(); B2.reset();
                for (uint32_t t : left) B1.grow(sbox[t]);
                for (uint32_t t : right) B2.grow(sbox[t]);
                for (const Cut &c : cuts) { B1.grow(c.lo_b); B2.grow(c.hi_b); }
                float N1 = (float)(left.size() + cuts.size()), N2 = (float)(right.size() + cuts.size());
                size_t n_cut = 0;
                for (const Cut &c : cuts) {
                    Box U1 = B1, U2 = B2; U1.grow(sbox[c.t]); U2.grow(sbox[c.t]);
                    const float c_split = B1.area() * N1 + B2.area() * N2;
                    const float c_left = U1.area() * N1 + B2.area() * (N2 - 1.0f);
                    const float c_right = B1.area() * (N1 - 1.0f) + U2.area() * N2;
                    if (kUnsplit && c_left < c_split && c_left <= c_right && N2 > 1.0f) { left.push_back(c.t); B1 = U1; N2 -= 1.0f; }
                    else if (kUnsplit && c_right < c_split && N1 > 1.0f) { right.push_back(c.t); B2 = U2; N1 -= 1.0f; }
                    else {
                        const uint32_t tri = items[c.t].first_tri;
                        set_box(c.t, c.lo_b); left.push_back(c.t);
                        right.push_back(add_reference(tri, c.hi_b));
                        ++n_cut;
                    }
                }
                *dup_budget -= (long long)n_cut;
            }
        }
        if (left.empty() || right.empty()) {
            left.clear(); right.clear();
            if (best_axis >= 0) for (uint32_t t : refs) (object_side(t) ? left : right).push_back(t);
            if (left.empty() || right.empty()) {
                if (!special && n_tris <= kLeafMax) return make_leaf(refs);
                left.assign(refs.begin(), refs.begin() + count / 2);               // degenerate: split by index
                right.assign(refs.begin() + count / 2, refs.end());
            }
        }
        refs.clear(); refs.shrink_to_fit();
        const int l = build(std::move(left), depth + 1u);
        const int r = build(std::move(right), depth + 1u);
        Tmp t; t.box = S.tmp[l].box; t.box.grow(S.tmp[r].box); t.left = l; t.right = r;
        S.tmp.push_back(t);
        if (defer_grain) top_inner.push_back((int)S.tmp.size() - 1);
        return (int)S.tmp.size() - 1;
    }
};

void padded(const Box &b, float *lo, float *hi, float rel = 2e-5f)
{
    float m = 0.0f;
    for (int k = 0; k < 3; ++k) m = std::max(m, std::max(std::fabs(b.lo[k]), std::fabs(b.hi[k])));
    float pad = rel * (1.0f + m);        // culling must stay conservative under f32 rounding
    for (int k = 0; k < 3; ++k) { lo[k] = b.lo[k] - pad; hi[k] = b.hi[k] + pad; }
}
void set_child(Node &n, int c, const Box &b, int32_t ref)
{
    float lo[3], hi[3];
    padded(b, lo, hi);
    node_set_child(n, c, lo, hi, ref);
}
void set_empty_child(Node &n, int c, int32_t ref)
{
    float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
    node_set_child(n, c, lo, hi, ref);
}

} // namespace

bool mesh_is_affine_box(const float *verts, uint32_t first, const float *xf, uint32_t face_tris[12])
{
    uint32_t cnt[6] = { 0, 0, 0, 0, 0, 0 }, corners[12][3];
    for (uint32_t t = 0; t < 12; ++t) {
        for (int v = 0; v < 3; ++v) {
            const float *p = verts + 9 * (size_t)(first + t) + 3 * v;
            uint32_t id = 0;
            for (int r = 0; r < 3; ++r) {
                const double q = (double)xf[4 * r] * p[0] + (double)xf[4 * r + 1] * p[1] + (double)xf[4 * r + 2] * p[2] + (double)xf[4 * r + 3];
                if (!(std::fabs(std::fabs(q) - 1.0) <= 1e-3)) return false;          // every vertex is a corner of the cube
                id |= q > 0.0 ? 1u << r : 0u;
            }
            corners[t][v] = id;
        }
        const uint32_t a = corners[t][0], b = corners[t][1], c = corners[t][2];
        if (a == b || b == c || a == c) return false;
        const uint32_t same = ~((a ^ b) | (a ^ c)) & 7u;                              // axes on which the three corners agree
        if (same != 1u && same != 2u && same != 4u) return false;                     // exactly one: the triangle lies in a face
        const uint32_t axis = same == 1u ? 0u : (same == 2u ? 1u : 2u);
        const uint32_t f = 2u * axis + ((a >> axis) & 1u);
        if (cnt[f] == 2u) return false;
        face_tris[2u * f + cnt[f]++] = first + t;
    }
    for (uint32_t f = 0; f < 6; ++f) {
        if (cnt[f] != 2u) return false;
        const uint32_t *A = corners[face_tris[2 * f] - first], *B = corners[face_tris[2 * f + 1] - first];
        uint32_t common[3], n = 0;
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) if (A[i] == B[j] && n < 3) common[n++] = A[i];
        // the two halves of the face: they share exactly its diagonal (two corners that differ on both free axes)
        if (n != 2u || (common[0] ^ common[1]) != (7u ^ (1u << (f >> 1)))) return false;
    }
    return true;
}

void build_bvh(const float *verts, uint32_t n, const BvhPrims *prims, BvhBuild &out)
{
    out.nodes.clear(); out.order.clear(); out.max_depth = 0; out.n_leaves = 0; out.packet_object.clear();
    if (n == 0) return;
    Shared S; S.verts = verts;
    Builder B(S); B.prims = prims;
    // leaves: two triangles for scenes staged in LDS (k_fused: 1 / 3 / 4 measured worse) and for the large scenes built with spatial
    // splits, four for scenes walked in HBM without them
    // The two cases cannot overlap: a slot costs 120 B of LDS (TriShade + half a TriPair) and the staged scene is limited to 64 KB
    // (fused_plan / wf_plan), so no scene above 546 triangles is ever walked in LDS — the choice is made on exactly that bound.
    // (The quantised HBM trees are still built for small scenes — a handful of nodes — because k_nlos_prepare and the
    // HBM instantiations requested explicitly walk them.)
    const bool never_in_lds = (size_t)n * (sizeof(TriShade) + sizeof(TriPair) / 2) > 64u * 1024u;
    // (before the spatial splits four triangles per leaf were best for the large scenes — config 5 at 256 spp, 1 / 2 / 3 / 4: 335 / 288 /
    // 283 / 275 ms; with them 2 / 3 / 4: k_wf_trace 110.0 / 116.1 / 113.9 ms)
    const bool sbvh = n >= 1024 && !mtr::knob("MTR_BVH_NO_SBVH");
    // (round 6, with the planned 8-wide collapse: leaf target 1 / 2 / 3 k_wf_trace 101.8 / 103.7 / 111.1 ms at 256 spp — the planner fills the nodes,
    // so single-triangle leaves where the SAH wants them cost no node steps any more)
    if (never_in_lds) { B.kLeafTarget = sbvh ? 1 : 4; B.kBins = 32; }          // (SAH bins 16 / 32 / 64: k_wf_trace 119.9 / 113.9 / 113.5 ms)
    if (const char *e = mtr::knob("MTR_BVH_BINS")) { B.kBins = atoi(e); if (B.kBins < 4) B.kBins = 4; if (B.kBins > Builder::kMaxBins) B.kBins = Builder::kMaxBins; }   // experiments
    if (const char *e = mtr::knob("MTR_BVH_LEAF")) { B.kLeafTarget = (uint32_t)atoi(e); if (B.kLeafTarget < 1) B.kLeafTarget = 1; if (B.kLeafTarget > 4) B.kLeafTarget = 4; }   // experiments
    S.tmp.reserve(3 * (size_t)n);
    for (uint32_t i = 0; i < n;) {
        const uint8_t kind = prims && prims->kind ? prims->kind[i] : 0;
        const int32_t obj = prims && prims->object ? prims->object[i] : -1;
        if (kind == 1 && i + 1 < n) { B.add_item(Item{ i, 2u, kItemQuad, -1 }, nullptr); i += 2; continue; }
        if (obj >= 0) {
            uint32_t j = i;
            while (j < n && prims->object[j] == obj) ++j;
            B.add_item(Item{ i, j - i, kItemObject, obj }, nullptr);
            i = j; continue;
        }
        B.add_item(Item{ i, 1u, kItemTri, -1 }, nullptr);
        ++i;
    }
    // (experiments only — MTR_BVH_ESC — since the builder splits spatially itself, below)
    // EARLY SPLIT CLIPPING of large mesh triangles (Ernst & Greiner 2007): a wall or floor triangle spanning the room has
    // a box that overlaps everything below it in the tree; it enters the build as several REFERENCES, each with the box of
    // the triangle clipped to one cell of a recursive midpoint split.  Intersection is unchanged (a leaf tests the whole
    // triangle, ties go to the original index, duplicates in one leaf are dropped), only culling gets tighter.
    // Large scenes only: the scenes staged in LDS are dominated by rectangles and object nodes.
    if (n >= 1024 && mtr::knob("MTR_BVH_ESC")) {
        Box scene; scene.reset();
        for (const Box &b : B.wbox) scene.grow(b);
        double frac = 1e-4;                                           // staircase (config 5 at 256 spp): off 296, 2e-3 287, 5e-4 290, 1e-4 283, 2e-5 290 ms
        if (const char *e = mtr::knob("MTR_BVH_SPLIT_FRAC")) frac = atof(e);
        const float a_max = scene.area() * (float)frac;
        size_t budget = n / 4;                                        // at most 25 % more references
        struct Piece { uint32_t item; std::vector<double> poly; };
        std::vector<Piece> work;
        for (uint32_t it = 0; it < (uint32_t)B.items.size(); ++it)
            if (B.items[it].type == kItemTri && B.wbox[it].area() > a_max) {
                const float *v = verts + 9 * (size_t)B.items[it].first_tri;
                Piece p; p.item = it; p.poly.assign(v, v + 9);
                work.push_back(std::move(p));
            }
        auto poly_box = [](const std::vector<double> &poly) { return mtr::poly_box((const double (*)[3])poly.data(), (int)(poly.size() / 3)); };
        auto smaller = [&](const Piece &x, const Piece &y) { return B.wbox[x.item].area() < B.wbox[y.item].area(); };
        std::make_heap(work.begin(), work.end(), smaller);            // largest piece first
        while (!work.empty() && budget > 0) {
            std::pop_heap(work.begin(), work.end(), smaller);
            Piece cur = std::move(work.back()); work.pop_back();
            const Box cb = B.wbox[cur.item];
            if (!(cb.area() > a_max)) continue;
            int axis = 0;
            for (int k = 1; k < 3; ++k) if (cb.hi[k] - cb.lo[k] > cb.hi[axis] - cb.lo[axis]) axis = k;
            const double pos = 0.5 * ((double)cb.lo[axis] + (double)cb.hi[axis]);
            const int np_ = (int)cur.poly.size() / 3;
            double in[16][3], lo_p[16][3], hi_p[16][3];
            if (np_ > 12) continue;
            for (int k = 0; k < np_; ++k) for (int c = 0; c < 3; ++c) in[k][c] = cur.poly[3 * k + c];
            const int nl = clip_poly(in, np_, axis, pos, true, lo_p), nh = clip_poly(in, np_, axis, pos, false, hi_p);
            if (nl < 3 || nh < 3) continue;                            // the plane misses the piece (degenerate): keep it whole
            Piece a, b2;
            a.item = cur.item; a.poly.assign(&lo_p[0][0], &lo_p[0][0] + 3 * nl);
            Box ba = poly_box(a.poly), bb;
            b2.poly.assign(&hi_p[0][0], &hi_p[0][0] + 3 * nh);
            bb = poly_box(b2.poly);
            // never grow beyond the piece being split (clipping can only shrink; the padding must not escape it either)
            for (int k = 0; k < 3; ++k) { ba.lo[k] = std::max(ba.lo[k], cb.lo[k]); ba.hi[k] = std::min(ba.hi[k], cb.hi[k]); bb.lo[k] = std::max(bb.lo[k], cb.lo[k]); bb.hi[k] = std::min(bb.hi[k], cb.hi[k]); }
            B.sbox[cur.item] = ba; B.wbox[cur.item] = ba;
            for (int k = 0; k < 3; ++k) B.cent[3 * (size_t)cur.item + k] = 0.5f * (ba.lo[k] + ba.hi[k]);
            B.add_reference(B.items[cur.item].first_tri, bb);
            b2.item = (uint32_t)B.items.size() - 1u;
            --budget;
            work.push_back(std::move(a)); std::push_heap(work.begin(), work.end(), smaller);
            work.push_back(std::move(b2)); std::push_heap(work.begin(), work.end(), smaller);
        }
    }
    // SPATIAL SPLITS during the build (the builder's comment; round 5): large scenes walked in HBM only; alpha decides how many
    // references are duplicated, the budget (as many duplicates as triangles) is a bound on memory, not a tuning knob.
    // Config 5 at 256 spp, k_wf_trace per render: no splits 161 ms, early split clipping (rounds 2-5) 131 ms, spatial splits 113.5 ms
    // (both together 121 ms: pieces cut before the build take the planes the builder would have chosen).
    // (round 6: alpha 3e-7 with room for 2 n duplicates — the parallel build pays for them: staircase build_bvh 0.98 s, 276 k leaves instead of 208 k;
    // alpha 1e-6 / 3e-7 / 1e-7 with leaf target 1: k_wf_trace 101.8 / 100.8 / 99.5 ms at 256 spp, the last at 342 k leaves and 1.2 s)
    long long dup_budget = sbvh ? 2ll * (long long)n : 0;
    if (sbvh) B.kAlpha = 3e-7f;
    if (const char *e = mtr::knob("MTR_BVH_SBVH_BUDGET")) dup_budget = (long long)(atof(e) * (double)n);
    if (const char *e = mtr::knob("MTR_BVH_SBVH_ALPHA")) B.kAlpha = (float)atof(e);
    if (mtr::knob("MTR_BVH_UNSPLIT")) B.kUnsplit = true;
    if (const char *e = mtr::knob("MTR_BVH_DEPTH_BUDGET")) B.kDepthBudget = (uint32_t)std::max(4, atoi(e));
    B.kPairCost = sbvh && !mtr::knob("MTR_BVH_NO_PAIR_COST");
    if (const char *e = mtr::knob("MTR_BVH_SBVH_BINS")) B.kSpatialBins = std::min(Builder::kMaxSpatialBins, std::max(4, atoi(e)));
    if (dup_budget) {
        Box scene; scene.reset();
        for (const Box &bx : B.wbox) scene.grow(bx);
        B.spatial = true; B.dup_budget = &dup_budget; B.root_area = scene.area();
    }
    const long long dup_budget0 = dup_budget; const size_t n_refs0 = B.items.size();
    // large scenes: subtrees of <= n / 64 references are built by worker threads (Builder::build_parallel_finish)
    B.n_threads = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    if (const char *e = mtr::knob("MTR_BVH_THREADS")) B.n_threads = (unsigned)std::max(1, atoi(e));       // experiments / tests
    if (n >= 16384) B.defer_grain = std::max<size_t>(2048, B.items.size() / 32);      // (whatever the number of threads: ONE algorithm, one tree)
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_b0 = now();
    const int root = B.build(B.order);
    const double t_b1 = now();
    if (B.defer_grain) B.build_parallel_finish();
    const double t_b2 = now();
    if (mtr::knob("MTR_BVH_VERBOSE")) fprintf(stderr, "build_bvh: top of the tree %.3f s, subtrees + stitching %.3f s\n", t_b1 - t_b0, t_b2 - t_b1);
    if (mtr::knob("MTR_BVH_VERBOSE"))
        fprintf(stderr, "build_bvh: %u triangles, %zu references before the build (early split clipping), %lld duplicated by spatial splits (budget %lld), %u threads\n",
                n, n_refs0, dup_budget0 - dup_budget, dup_budget0, B.n_threads);

    // flatten: one packet per inner Tmp node
    // leaves are laid out in slot space: each starts on an even slot, odd leaves get a pad slot
    auto leaf_ref = [&](const Tmp &t) -> int32_t {
        const uint32_t first = (uint32_t)out.order.size();
        for (uint32_t k = 0; k < t.count; ++k) out.order.push_back(S.leaf_tris[t.first + k]);
        if (t.count & 1u) out.order.push_back(kPadSlot);
        uint32_t code = (first << 2) | (t.count - 1u);
        if (t.quad) code |= kLeafQuadBit;
        return (int32_t)~code;
    };
    struct Entry { int tmp; int packet; uint32_t depth; };
    std::vector<Entry> stack;
    const Tmp &R = S.tmp[root];
    if (R.left < 0) {                        // the whole scene is one leaf
        Node nd{};
        const int32_t ref = leaf_ref(R);
        set_child(nd, 0, R.box, ref); set_empty_child(nd, 1, ref);
        out.nodes.push_back(nd); out.packet_object.push_back(-1); out.max_depth = 1; out.n_leaves = 1;
    } else {
        out.nodes.emplace_back(); out.packet_object.push_back(R.object);
        stack.push_back({ root, 0, 1 });
        while (!stack.empty()) {
            Entry it = stack.back(); stack.pop_back();
            out.max_depth = std::max(out.max_depth, it.depth);
            const Tmp &t = S.tmp[it.tmp];
            const Tmp &L = S.tmp[t.left], &Rr = S.tmp[t.right];
            Node nd{};
            int32_t c0, c1;
            if (L.left < 0) { c0 = leaf_ref(L); out.n_leaves++; }
            else { c0 = (int32_t)out.nodes.size(); out.nodes.emplace_back(); out.packet_object.push_back(L.object); stack.push_back({ t.left, c0, it.depth + 1 }); }
            if (Rr.left < 0) { c1 = leaf_ref(Rr); out.n_leaves++; }
            else { c1 = (int32_t)out.nodes.size(); out.nodes.emplace_back(); out.packet_object.push_back(Rr.object); stack.push_back({ t.right, c1, it.depth + 1 }); }
            set_child(nd, 0, L.box, c0); set_child(nd, 1, Rr.box, c1);
            out.nodes[it.packet] = nd;
        }
    }
}

namespace {

struct WChild { float lo[3], hi[3]; int32_t ref; };

void packet_children(const Node &n, std::vector<WChild> &out)
{
    const float *f = &n.q[0].x;
    for (int c = 0; c < 2; ++c) {
        WChild w;
        for (int k = 0; k < 3; ++k) { w.lo[k] = f[4 * k + c]; w.hi[k] = f[4 * k + 2 + c]; }
        w.ref = (int32_t)fbits(f[12 + c]);
        if (w.lo[0] <= w.hi[0]) out.push_back(w);          // an absent child has an inverted box
    }
}
float half_area(const WChild &w)
{
    const float dx = w.hi[0] - w.lo[0], dy = w.hi[1] - w.lo[1], dz = w.hi[2] - w.lo[2];
    return dx * dy + dy * dz + dz * dx;
}

bool is_quad_ref(int32_t ref) { return ref < 0 && ((~(uint32_t)ref) & kLeafQuadBit) != 0u; }

// every leaf below a BVH2 packet
void subtree_leaves(const BvhBuild &bvh, int32_t packet, std::vector<int32_t> &out)
{
    std::vector<WChild> ch;
    packet_children(bvh.nodes[packet], ch);
    for (const WChild &w : ch) { if (w.ref < 0) out.push_back(w.ref); else subtree_leaves(bvh, w.ref, out); }
}

// COLLAPSE PLAN (round 6; VERDICT r5 #3c: "the SAH that scores the 8-wide collapse"; Ylitie, Karras, Laine 2017, section 3.1).  The
// greedy collapse below opens the child of largest area until a node is full.  The plan instead MINIMISES the SAH cost of the wide
// tree over all ways of cutting the binary tree into 8-wide nodes, by dynamic programming over the binary tree's subtrees:
//   cost[S][1]     = S as ONE slot of its parent: a leaf as it is, or a wide node of its own = area(S) c_node + distribute(S, 8)
//   cost[S][i > 1] = S dissolved into at most i slots = min(cost[S][i - 1], distribute(S, i))
//   distribute(S, j) = min over k of cost[left(S)][k] + cost[right(S)][j - k]
// with c_node the price of a node step and c_leaf that of a pair of triangles tested (a subtree = a child slot of a BVH2 packet: 2 p + side).
struct CollapsePlan {
    static constexpr int kW = 8;
    std::vector<float> cost;                 // [slot][kW]: cost[s * kW + (i - 1)]
    std::vector<uint8_t> opaque;             // [packet]: an object subtree that becomes ONE object node of its own — never dissolved into its parent
    float c_node = 1.0f, c_leaf = 0.6f;
    float at(size_t s, int i) const { return cost[s * kW + (size_t)(i - 1)]; }
};
void plan_collapse(const BvhBuild &bvh, CollapsePlan &P)
{
    const size_t np = bvh.nodes.size();
    P.cost.assign(2 * np * CollapsePlan::kW, INFINITY);
    auto distribute = [&](size_t q, int j) {            // the children of packet q in at most j slots
        float best = INFINITY;
        for (int k = 1; k < j; ++k) best = std::min(best, P.at(2 * q, k) + P.at(2 * q + 1, j - k));
        return best;
    };
    for (size_t p = np; p-- > 0;) {                      // children have larger indices than their parents (build_bvh's flattening)
        const float *f = &bvh.nodes[p].q[0].x;
        for (int c = 0; c < 2; ++c) {
            const size_t s = 2 * p + (size_t)c;
            float lo[3], hi[3];
            for (int k = 0; k < 3; ++k) { lo[k] = f[4 * k + c]; hi[k] = f[4 * k + 2 + c]; }
            if (!(lo[0] <= hi[0])) { for (int i = 1; i <= CollapsePlan::kW; ++i) P.cost[s * CollapsePlan::kW + (size_t)(i - 1)] = 0.0f; continue; }      // an absent child costs nothing
            const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2], area = dx * dy + dy * dz + dz * dx;
            const int32_t ref = (int32_t)fbits(f[12 + c]);
            if (ref < 0) {
                const uint32_t code = ~(uint32_t)ref, cnt = (code & kLeafQuadBit) ? 2u : (code & 3u) + 1u;
                const float v = area * P.c_leaf * (float)((cnt + 1u) / 2u);
                for (int i = 1; i <= CollapsePlan::kW; ++i) P.cost[s * CollapsePlan::kW + (size_t)(i - 1)] = v;
                continue;
            }
            const size_t q = (size_t)ref;
            if (!P.opaque.empty() && P.opaque[q]) { for (int i = 1; i <= CollapsePlan::kW; ++i) P.cost[s * CollapsePlan::kW + (size_t)(i - 1)] = area * P.c_node; continue; }
            P.cost[s * CollapsePlan::kW] = area * P.c_node + distribute(q, CollapsePlan::kW);
            for (int i = 2; i <= CollapsePlan::kW; ++i) P.cost[s * CollapsePlan::kW + (size_t)(i - 1)] = std::min(P.at(s, i - 1), distribute(q, i));
        }
    }
}
// the slots the plan gives the children of packet q when they may take at most j: appended to `out`
void plan_children(const BvhBuild &bvh, const CollapsePlan &P, size_t q, int j, std::vector<WChild> &out)
{
    int best_k = 1; float best = INFINITY;
    for (int k = 1; k < j; ++k) { const float v = P.at(2 * q, k) + P.at(2 * q + 1, j - k); if (v < best) { best = v; best_k = k; } }
    std::vector<WChild> two;
    {
        const float *f = &bvh.nodes[q].q[0].x;
        for (int c = 0; c < 2; ++c) {
            WChild w;
            for (int k = 0; k < 3; ++k) { w.lo[k] = f[4 * k + c]; w.hi[k] = f[4 * k + 2 + c]; }
            w.ref = (int32_t)fbits(f[12 + c]);
            two.push_back(w);
        }
    }
    const int share[2] = { best_k, j - best_k };
    for (int c = 0; c < 2; ++c) {
        const WChild &w = two[c];
        if (!(w.lo[0] <= w.hi[0])) continue;                 // absent
        const size_t s = 2 * q + (size_t)c;
        int i = share[c];
        while (i > 1 && P.at(s, i - 1) <= P.at(s, i)) --i;   // the fewest slots that reach the same cost
        if (w.ref < 0 || i == 1 || (!P.opaque.empty() && P.opaque[(size_t)w.ref])) out.push_back(w);           // one slot: a leaf, or a wide node of its own
        else plan_children(bvh, P, (size_t)w.ref, i, out);   // dissolved into its parent
    }
}

template <uint32_t W>
uint32_t wide_rec(const BvhBuild &bvh, const BvhPrims *prims, const float *verts, int32_t packet, std::vector<WNodeT<W>> &wide,
                  uint32_t level, uint32_t &levels, size_t width, const CollapsePlan *plan = nullptr)
{
    levels = std::max(levels, level);
    const uint32_t me = (uint32_t)wide.size();
    wide.emplace_back();
    std::vector<WChild> ch;
    // an OBJECT subtree with few leaves becomes one node whose boxes live in the object's own space (8-wide tree only)
    const float *xf = nullptr;
    bool is_box = false;
    const bool objects_on = W == 8 && prims && prims->object_xf && verts && !mtr::knob("MTR_NO_OBJECT_NODES");
    if (objects_on) {
        const int32_t obj = bvh.packet_object[packet];
        std::vector<int32_t> leaves;
        if (obj >= 0) subtree_leaves(bvh, packet, leaves);
        if (obj >= 0 && leaves.size() <= width) {
            xf = prims->object_xf + 12 * (size_t)obj;
            for (int32_t ref : leaves) {
                const uint32_t code = ~(uint32_t)ref, first = code >> 2, cnt = (code & 3u) + 1u;
                Box b; b.reset();
                for (uint32_t k = 0; k < cnt; ++k) b.grow(tri_box(verts + 9 * (size_t)bvh.order[first + k], xf));
                WChild w; w.ref = ref;
                padded(b, w.lo, w.hi, 1e-4f);          // + the rounding of the ray's own transform in the kernel
                ch.push_back(w);
            }
            // a BOX: the object is an affine cube and its six leaves are its six faces -> children in face order
            if (leaves.size() == 6 && !mtr::knob("MTR_NO_BOX_NODES")) {
                uint32_t first = 0xffffffffu, faces[12];
                for (int32_t ref : leaves) {
                    const uint32_t code = ~(uint32_t)ref;
                    for (uint32_t k = 0; k <= (code & 3u); ++k) first = std::min(first, bvh.order[(code >> 2) + k]);
                }
                bool ok = mesh_is_affine_box(verts, first, xf, faces);
                std::vector<WChild> by_face(6);
                uint32_t seen = 0u;
                for (size_t c = 0; ok && c < 6; ++c) {
                    const uint32_t code = ~(uint32_t)ch[c].ref, slot = code >> 2;
                    ok = (code & 3u) == 1u;
                    for (uint32_t f = 0; ok && f < 6; ++f) {
                        const uint32_t a = bvh.order[slot], b = bvh.order[slot + 1];
                        if ((a == faces[2 * f] && b == faces[2 * f + 1]) || (b == faces[2 * f] && a == faces[2 * f + 1])) { by_face[f] = ch[c]; seen |= 1u << f; }
                    }
                }
                if (ok && seen == 63u) { ch = by_face; is_box = true; }
            }
        }
    }
    if (!xf && plan) plan_children(bvh, *plan, (size_t)packet, (int)W, ch);
    else if (!xf) {
        packet_children(bvh.nodes[packet], ch);
        // (an object subtree that will become an object node of its own is never dissolved into its parent)
        auto opaque = [&](int32_t ref) {
            if (!objects_on || bvh.packet_object[ref] < 0) return false;
            std::vector<int32_t> lv; subtree_leaves(bvh, ref, lv);
            return lv.size() <= width;
        };
        for (;;) {
            int best = -1; float best_a = -1.0f;
            for (size_t i = 0; i < ch.size(); ++i)
                if (ch[i].ref >= 0 && half_area(ch[i]) > best_a && !opaque(ch[i].ref)) { best = (int)i; best_a = half_area(ch[i]); }
            if (best < 0) break;
            std::vector<WChild> sub;
            packet_children(bvh.nodes[ch[best].ref], sub);
            if (ch.size() - 1 + sub.size() > width) break;
            ch.erase(ch.begin() + best);
            ch.insert(ch.end(), sub.begin(), sub.end());
        }
    }
    // rectangles first (8-wide tree: the walk always visits them first, wide_advance), the others in walk order: by centroid
    // along the axis on which their centroids spread most
    size_t nq = 0;
    if (W == 8 && !is_box) nq = (size_t)(std::stable_partition(ch.begin(), ch.end(), [](const WChild &w) { return is_quad_ref(w.ref); }) - ch.begin());
    float clo[3] = { INFINITY, INFINITY, INFINITY }, chi[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (size_t i = nq; i < ch.size(); ++i)
        for (int k = 0; k < 3; ++k) { const float c = 0.5f * (ch[i].lo[k] + ch[i].hi[k]); clo[k] = std::min(clo[k], c); chi[k] = std::max(chi[k], c); }
    int axis = 0;
    for (int k = 1; k < 3; ++k) if (chi[k] - clo[k] > chi[axis] - clo[axis]) axis = k;
    if (!is_box) std::stable_sort(ch.begin() + nq, ch.end(), [axis](const WChild &a, const WChild &b) { return a.lo[axis] + a.hi[axis] < b.lo[axis] + b.hi[axis]; });

    WNodeT<W> nd{};
    float *f = &nd.box[0].x;
    for (int c = 0; c < (int)W; ++c) {
        const int j = c >> 1, h = c & 1;
        for (int k = 0; k < 3; ++k) {
            f[4 * (3 * j + k) + h] = c < (int)ch.size() ? ch[c].lo[k] : INFINITY;
            f[4 * (3 * j + k) + 2 + h] = c < (int)ch.size() ? ch[c].hi[k] : -INFINITY;
        }
        nd.ref[c] = 0;
    }
    nd.axis = (uint32_t)axis; nd.count = (uint32_t)ch.size(); nd.n_quads = (uint32_t)nq; nd.flags = (xf ? 1u : 0u) | (is_box ? 2u : 0u);
    for (int k = 0; k < 12; ++k) nd.xf[k] = xf ? xf[k] : 0.0f;
    for (size_t c = 0; c < ch.size(); ++c)
        nd.ref[c] = ch[c].ref >= 0 ? (int32_t)wide_rec<W>(bvh, prims, verts, ch[c].ref, wide, level + 1, levels, width, plan) : ch[c].ref;
    wide[me] = nd;
    return me;
}

} // namespace

uint32_t build_wide(const BvhBuild &bvh, const BvhPrims *prims, const float *verts, std::vector<WNode> &wide)
{
    wide.clear();
    if (bvh.nodes.empty()) return 0;
    size_t width = kWide;
    if (const char *e = mtr::knob("MTR_WIDE_WIDTH")) { int w = atoi(e); width = w < 2 ? 2 : (w > (int)kWide ? (int)kWide : w); }   // experiments
    uint32_t levels = 0;
    // the collapse plan for the 8-wide tree of the scenes staged in LDS too; object subtrees stay whole (MTR_BVH_GREEDY_LDS in the experiments
    // build = the greedy collapse of rounds 2 - 5).  Config 2's film over the Cornell box with 48-triangle boxes (108 triangles): wavefront
    // organisation 122.3 -> 113.3 ms, fused (forced) 208.9 -> 147.4 ms; the Cornell box itself (one root, two box nodes) and config 4: unchanged.
    CollapsePlan plan;
    const bool planned = kWide == 8 && width == 8 && mtr::knob("MTR_BVH_GREEDY_LDS") == nullptr;
    if (planned) {
        plan.opaque.assign(bvh.nodes.size(), 0);
        const bool objects_on = prims && prims->object_xf && verts && !mtr::knob("MTR_NO_OBJECT_NODES");
        for (size_t q = 0; objects_on && q < bvh.nodes.size(); ++q)
            if (bvh.packet_object[q] >= 0) { std::vector<int32_t> lv; subtree_leaves(bvh, (int32_t)q, lv); plan.opaque[q] = lv.size() <= width; }
        plan_collapse(bvh, plan);
    }
    wide_rec<kWide>(bvh, prims, verts, 0, wide, 1, levels, width, planned && !plan.opaque[0] ? &plan : nullptr);
    if (mtr::knob("MTR_BVH_VERBOSE")) fprintf(stderr, "build_wide: %zu nodes, %u levels%s\n", wide.size(), levels, planned ? " (planned collapse)" : "");
    return levels;
}
namespace {

// collapse to width W, breadth-first order (the top of the tree — the nodes every ray visits — sits at the lowest indices),
// then the children's planes quantised to 8 bits on the node's own grid: origin = the node's lower corner, one power-of-two
// step per axis, the smallest whose 255 multiples cover the extent; every plane rounds outwards (checked in f64).
// emit(i, node, meta, plo[3], phi[3]): plo[k] / phi[k] hold byte c = child c of axis k (64-bit: up to 8 children)
template <uint32_t W, class Emit>
uint32_t build_quantised(const BvhBuild &bvh, size_t &n_out, Emit emit)
{
    uint32_t levels = 0;
    std::vector<WNodeT<W>> full;
    CollapsePlan plan;
    // (the quantised 8-wide tree of the scenes walked in HBM; MTR_BVH_GREEDY8 in the experiments build = the greedy collapse of rounds 3 - 5.
    // Staircase: 65 957 nodes of 4.15 children -> 42 081 of 5.94, sum of node areas 2982 -> 2713; config 5 at 256 spp k_wf_trace 109.3 -> 104.1 ms,
    // the same with a leaf price of 0.3 / 0.6 / 1.0 node steps)
    const bool planned = W == 8 && mtr::knob("MTR_BVH_GREEDY8") == nullptr;
    if (planned) {
        if (const char *e = mtr::knob("MTR_BVH_PLAN8_LEAF")) plan.c_leaf = (float)atof(e);
        plan_collapse(bvh, plan);
    }
    wide_rec<W>(bvh, nullptr, nullptr, 0, full, 1, levels, W, planned ? &plan : nullptr);
    if (mtr::knob("MTR_BVH_VERBOSE")) {
        double sah = 0.0; size_t n_children = 0;
        for (const WNodeT<W> &w : full) {
            const float *f = &w.box[0].x;
            Box b; b.reset();
            for (uint32_t c = 0; c < w.count; ++c) { float lo[3], hi[3]; for (int k = 0; k < 3; ++k) { lo[k] = f[4 * (3 * (c >> 1) + k) + (c & 1)]; hi[k] = f[4 * (3 * (c >> 1) + k) + 2 + (c & 1)]; } b.grow(lo); b.grow(hi); }
            sah += (double)b.area(); n_children += w.count;
        }
        fprintf(stderr, "build_quantised<%u>: %zu nodes, %.2f children per node, %u levels, sum of node areas %.4g%s\n", W, full.size(), (double)n_children / (double)full.size(), levels, sah, planned ? " (planned collapse)" : "");
    }
    {
        std::vector<uint32_t> order{ 0u }, pos(full.size(), 0u);
        for (size_t i = 0; i < order.size(); ++i)
            for (uint32_t c = 0; c < full[order[i]].count; ++c)
                if (full[order[i]].ref[c] >= 0) order.push_back((uint32_t)full[order[i]].ref[c]);
        for (size_t i = 0; i < order.size(); ++i) pos[order[i]] = (uint32_t)i;
        std::vector<WNodeT<W>> bfs(full.size());
        for (size_t i = 0; i < order.size(); ++i) {
            bfs[i] = full[order[i]];
            for (uint32_t c = 0; c < bfs[i].count; ++c) if (bfs[i].ref[c] >= 0) bfs[i].ref[c] = (int32_t)pos[bfs[i].ref[c]];
        }
        full.swap(bfs);
    }
    n_out = full.size();
    for (size_t i = 0; i < full.size(); ++i) {
        const WNodeT<W> &w = full[i];
        const float *f = &w.box[0].x;
        auto lo = [&](uint32_t c, int k) { return f[4 * (3 * (c >> 1) + k) + (c & 1)]; };
        auto hi = [&](uint32_t c, int k) { return f[4 * (3 * (c >> 1) + k) + 2 + (c & 1)]; };
        float org[3];
        uint32_t meta = (w.axis << 24) | (w.count << 26);
        uint64_t plo[3] = { 0, 0, 0 }, phi[3] = { 0, 0, 0 };
        for (int k = 0; k < 3; ++k) {
            float o = INFINITY, top = -INFINITY;
            for (uint32_t c = 0; c < w.count; ++c) { o = std::min(o, lo(c, k)); top = std::max(top, hi(c, k)); }
            org[k] = o;
            int e = -100;
            const double ext = (double)top - (double)o;
            if (ext > 0.0) e = std::max(-100, (int)std::ceil(std::log2(ext / 255.0)));
            for (;; ++e) {
                const double step = std::ldexp(1.0, e);
                bool ok = true;
                uint64_t wl = 0, wh = 0;
                for (uint32_t c = 0; c < w.count && ok; ++c) {
                    long ql = (long)std::floor(((double)lo(c, k) - (double)o) / step), qh = (long)std::ceil(((double)hi(c, k) - (double)o) / step);
                    while (ql > 0 && (double)o + (double)ql * step > (double)lo(c, k)) --ql;
                    while ((double)o + (double)qh * step < (double)hi(c, k)) ++qh;
                    if (ql < 0) ql = 0;
                    if (qh > 255) { ok = false; break; }
                    wl |= (uint64_t)ql << (8 * c); wh |= (uint64_t)qh << (8 * c);
                }
                if (ok) { plo[k] = wl; phi[k] = wh; break; }
            }
            meta |= (uint32_t)(e + 127) << (8 * k);
        }
        emit(i, w, org, meta, plo, phi);
    }
    return levels;
}

} // namespace

uint32_t build_wide4(const BvhBuild &bvh, std::vector<QNode4> &wide)
{
    wide.clear();
    if (bvh.nodes.empty()) return 0;
    size_t n = 0;
    std::vector<QNode4> out;
    const uint32_t levels = build_quantised<4>(bvh, n, [&](size_t i, const WNodeT<4> &w, const float *org, uint32_t meta, const uint64_t *plo, const uint64_t *phi) {
        if (out.size() <= i) out.resize(i + 1);
        QNode4 q{};
        q.q[0] = q4{ org[0], org[1], org[2], bitsf(meta) };
        for (int c = 0; c < 4; ++c) (&q.q[1].x)[c] = bitsf((uint32_t)w.ref[c]);
        q.q[2] = q4{ bitsf((uint32_t)plo[0]), bitsf((uint32_t)plo[1]), bitsf((uint32_t)plo[2]), bitsf((uint32_t)phi[0]) };
        q.q[3] = q4{ bitsf((uint32_t)phi[1]), bitsf((uint32_t)phi[2]), 0.0f, 0.0f };
        out[i] = q;
    });
    wide.swap(out);
    return levels;
}

uint32_t build_wide8q(const BvhBuild &bvh, std::vector<QNode8> &wide)
{
    wide.clear();
    if (bvh.nodes.empty()) return 0;
    size_t n = 0;
    std::vector<QNode8> out;
    const uint32_t levels = build_quantised<8>(bvh, n, [&](size_t i, const WNodeT<8> &w, const float *org, uint32_t meta, const uint64_t *plo, const uint64_t *phi) {
        if (out.size() <= i) out.resize(i + 1);
        QNode8 q{};
        auto lo32 = [](uint64_t v) { return bitsf((uint32_t)v); };
        auto hi32 = [](uint64_t v) { return bitsf((uint32_t)(v >> 32)); };
        q.q[0] = q4{ org[0], org[1], org[2], bitsf(meta) };
        q.q[1] = q4{ lo32(plo[0]), hi32(plo[0]), lo32(plo[1]), hi32(plo[1]) };
        q.q[2] = q4{ lo32(plo[2]), hi32(plo[2]), lo32(phi[0]), hi32(phi[0]) };
        q.q[3] = q4{ lo32(phi[1]), hi32(phi[1]), lo32(phi[2]), hi32(phi[2]) };
        for (int c = 0; c < 8; ++c) (&q.q[4].x)[c] = bitsf((uint32_t)w.ref[c]);
        out[i] = q;
    });
    wide.swap(out);
    return levels;
}

} // namespace mtr
