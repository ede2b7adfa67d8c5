// mtr_bvh.h — host-side BVH2 builder (binned SAH) producing 64-byte node packets.
#pragma once
#include <vector>
#include <stdint.h>
#include "mtr_core.h"

namespace mtr {

struct BvhBuild {
    std::vector<Node> nodes;        // packet 0 is the root
    std::vector<uint32_t> order;    // order[i] = original index of the triangle stored at slot i
    uint32_t max_depth = 0;         // packets on the longest root-to-leaf chain (= traversal stack bound)
    uint32_t n_leaves = 0;
};

// verts: n*9 floats (p0 p1 p2 per triangle, world space)
void build_bvh(const float *verts, uint32_t n, BvhBuild &out);

} // namespace mtr
