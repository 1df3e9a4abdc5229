// GPU-side LBVH builder for one triangle mesh (pt_bvh_device.hip).
#pragma once
#include <hip/hip_runtime_api.h>

#include <vector>

#include "pt_types.h"

namespace pt {
struct DeviceBlas {
    std::vector<BvhNode> nodes;     // node 0 = root; child references are LOCAL: REF_NODE | index into `nodes`,
                                    // REF_TRIS | (count-1) << 27 | first position in `order`
    std::vector<uint32_t> order;    // BLAS (leaf) order: position -> face index of the mesh
    int depth = 0;                  // deepest leaf, counted like the host builder's depth_reached
};
// false: nothing built (too few triangles, a tree deeper than max_depth, or a HIP error) — build on the host instead
// median_below: subtrees of at most this many triangles are split in the middle of their (Morton-sorted) run instead of at the
// code's highest differing bit — balanced bottoms leave the depth budget to the upper levels, where the Morton splits matter
bool build_blas_device(const TriD* tris, uint32_t n, const double mesh_lo[3], const double mesh_hi[3], uint32_t leaf_max, uint32_t max_depth,
                       uint32_t median_below, DeviceBlas& out, hipStream_t st);
}  // namespace pt
