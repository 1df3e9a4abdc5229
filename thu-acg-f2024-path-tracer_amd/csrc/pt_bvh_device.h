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
bool build_blas_device(const TriD* tris, uint32_t n, const double mesh_lo[3], const double mesh_hi[3], uint32_t leaf_max, uint32_t max_depth,
                       DeviceBlas& out, hipStream_t st);
}  // namespace pt
