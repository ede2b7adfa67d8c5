// mtr_bvh.h — host-side BVH2 builder (binned SAH) producing 64-byte node packets.
#pragma once
#include <vector>
#include <stdint.h>
#include "mtr_core.h"

namespace mtr {

constexpr uint32_t kPadSlot = 0xffffffffu;

struct BvhBuild {
    std::vector<Node> nodes;        // packet 0 is the root
    // order[slot] = original index of the triangle stored at that slot; every leaf starts on an EVEN slot and a leaf with an
    // odd triangle count is followed by one pad slot (kPadSlot)
    std::vector<uint32_t> order;
    uint32_t max_depth = 0;         // packets on the longest root-to-leaf chain (= traversal stack bound)
    uint32_t n_leaves = 0;
};

// verts: n*9 floats (p0 p1 p2 per triangle, world space)
void build_bvh(const float *verts, uint32_t n, BvhBuild &out);

// Collapses the BVH2 into a tree of up to 8-wide nodes (WNode, mtr_core.h) over the SAME leaves and (padded) boxes:
// starting from a packet's two children, the inner child with the largest surface area is replaced by its own children
// until eight are held or only leaves remain.  wide[0] is the root; returns the number of levels.
uint32_t build_wide(const BvhBuild &bvh, std::vector<WNode> &wide);
// the same, 4 wide (one node per 128-byte line): scenes walked in HBM
uint32_t build_wide4(const BvhBuild &bvh, std::vector<QNode4> &wide);

} // namespace mtr
