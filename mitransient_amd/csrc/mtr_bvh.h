// mtr_bvh.h — host-side BVH2 builder (binned SAH) producing 64-byte node packets.
#pragma once
#include <vector>
#include <stdint.h>
#include "mtr_core.h"

namespace mtr {

constexpr uint32_t kPadSlot = 0xffffffffu;

// optional per-triangle annotations of the build (all arrays may be null)
struct BvhPrims {
    const uint8_t *kind = nullptr;      // [n] 0 mesh triangle, 1 first / 2 second carrier triangle of an analytic rectangle
    const int32_t *object = nullptr;    // [n] index of the OBJECT (small mesh shape with a known transform) the triangle belongs to, or -1
    const float *object_xf = nullptr;   // [n_objects][12] world -> object affine map (rows R | T)
};

struct BvhBuild {
    std::vector<Node> nodes;        // packet 0 is the root
    // order[slot] = original index of the triangle stored at that slot; every leaf starts on an EVEN slot and a leaf with an
    // odd triangle count is followed by one pad slot (kPadSlot)
    std::vector<uint32_t> order;
    uint32_t max_depth = 0;         // packets on the longest root-to-leaf chain (= traversal stack bound)
    uint32_t n_leaves = 0;
    std::vector<int32_t> packet_object;   // per packet: the object whose subtree starts there, or -1
};

// Is the mesh [first, first + 12) the image of the cube [-1,1]^3 under the inverse of `xf` (world -> object rows), every
// face split into two triangles along a diagonal?  face_tris[2 f], [2 f + 1] = the triangles of face f = 2 * axis +
// (object coordinate = +1), lower index first.
bool mesh_is_affine_box(const float *verts, uint32_t first, const float *xf, uint32_t face_tris[12]);

// verts: n*9 floats (p0 p1 p2 per triangle, world space)
void build_bvh(const float *verts, uint32_t n, const BvhPrims *prims, BvhBuild &out);

// Collapses the BVH2 into a tree of up to 8-wide nodes (WNode, mtr_core.h) over the SAME leaves and (padded) boxes:
// starting from a packet's two children, the inner child with the largest surface area is replaced by its own children
// until eight are held or only leaves remain.  wide[0] is the root; returns the number of levels.
// Object subtrees with at most eight leaves become object nodes (boxes in the object's space, mtr_core.h WNodeT) — BOX nodes
// when the object is an affine cube whose six leaves are its six faces (mtr_core.h box_select); rectangle children are
// stored first.
uint32_t build_wide(const BvhBuild &bvh, const BvhPrims *prims, const float *verts, std::vector<WNode> &wide);
// the same, 4 wide (one node per 128-byte line): scenes walked in HBM
uint32_t build_wide4(const BvhBuild &bvh, std::vector<QNode4> &wide);
// ... and 8 wide with the same quantisation (QNode8, 96 bytes)
uint32_t build_wide8q(const BvhBuild &bvh, std::vector<QNode8> &wide);

} // namespace mtr
