// GPU-side BVH builder for triangle meshes (the reference's builder is the O(n^2) sweep SAH of
// bvh.rs:24-121, the default builder here a host binned SAH — pt_scene.cpp). LBVH after Karras 2012: 63-bit Morton codes
// of the triangle centroids, one radix sort, the radix tree built in parallel (one thread per internal node), boxes
// fitted bottom-up with one atomic arrival counter per node, then subtrees of <= leaf_max triangles collapsed into the
// triangle leaves of pt_types.h and the remaining internal nodes compacted into BvhNode records.
//
// The closest hit does not depend on the tree (minimum t, ties -> larger primitive id), so a mesh built
// here renders bit for bit like one built on the host (tests/test_gpu_parity.py::test_device_bvh_builder_bit_exact). An
// LBVH is shallower in build time and worse in traversal cost than a SAH tree, so it is meant for meshes large enough
// that the host build time matters (pt_world_set_device_bvh_threshold); the traversal kernels' LDS stack bounds the
// depth a tree may have, and a tree that comes out deeper is rejected (the caller falls back to the host builder).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>

#include "pt_bvh_device.h"

namespace pt {
namespace {

constexpr int BT = 256;
constexpr uint32_t LEAF_BIT = 0x80000000u;

struct Box64 {
    double lo[3], hi[3];
};

__device__ inline uint64_t spread21(uint64_t v) {   // 21 bits -> every third bit
    v &= 0x1FFFFFull;
    v = (v | (v << 32)) & 0x1F00000000FFFFull;
    v = (v | (v << 16)) & 0x1F0000FF0000FFull;
    v = (v | (v << 8)) & 0x100F00F00F00F00Full;
    v = (v | (v << 4)) & 0x10C30C30C30C30C3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}

__global__ void k_keys(const TriD* tris, uint32_t n, Box64 mesh, Box64* tri_box, uint64_t* keys, uint32_t* vals) {
    const uint32_t i = blockIdx.x * BT + threadIdx.x;
    if (i >= n) return;
    const TriD t = tris[i];
    Box64 b;
    uint64_t code = 0;
    for (int a = 0; a < 3; ++a) {
        b.lo[a] = fmin(t.v0[a], fmin(t.v1[a], t.v2[a]));
        b.hi[a] = fmax(t.v0[a], fmax(t.v1[a], t.v2[a]));
        const double ext = mesh.hi[a] - mesh.lo[a];
        double x = ext > 0.0 ? ((b.lo[a] + b.hi[a]) * 0.5 - mesh.lo[a]) / ext : 0.0;
        x = fmin(fmax(x, 0.0), 1.0);
        uint64_t q = (uint64_t)(x * 2097152.0);
        if (q > 2097151ull) q = 2097151ull;
        code |= spread21(q) << a;
    }
    tri_box[i] = b;
    keys[i] = code;
    vals[i] = i;
}

// common-prefix length of the keys at sorted positions i and j; equal keys are told apart by the position itself
__device__ inline int delta(const uint64_t* keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const uint64_t a = keys[i], b = keys[j];
    if (a != b) return __clzll((long long)(a ^ b));
    return 64 + __clz((int)((uint32_t)i ^ (uint32_t)j));
}

// Karras 2012, one thread per internal node i in [0, n-1): its key range and its two children
__global__ void k_hierarchy(const uint64_t* keys, int n, uint32_t* child /* 2 per node */, uint32_t* parent_node /* n-1 */, uint32_t* parent_leaf /* n */,
                            uint32_t* range /* first, last per node */) {
    const int i = (int)(blockIdx.x * BT + threadIdx.x);
    if (i >= n - 1) return;
    const int d = delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = delta(keys, n, i, i - d);
    int lmax = 2;
    while (delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(keys, n, i, j);
    int s = 0;
    for (int t = (l + 1) / 2;; t = (t + 1) / 2) {
        if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
        if (t == 1) break;
    }
    const int gamma = i + s * d + (d < 0 ? -1 : 0);
    const int first = i < j ? i : j, last = i < j ? j : i;
    const uint32_t left = first == gamma ? ((uint32_t)gamma | LEAF_BIT) : (uint32_t)gamma;
    const uint32_t right = last == gamma + 1 ? ((uint32_t)(gamma + 1) | LEAF_BIT) : (uint32_t)(gamma + 1);
    child[2 * i] = left;
    child[2 * i + 1] = right;
    range[2 * i] = (uint32_t)first;
    range[2 * i + 1] = (uint32_t)last;
    if (left & LEAF_BIT) parent_leaf[gamma] = (uint32_t)i; else parent_node[gamma] = (uint32_t)i;
    if (right & LEAF_BIT) parent_leaf[gamma + 1] = (uint32_t)i; else parent_node[gamma + 1] = (uint32_t)i;
}

__device__ inline Box64 merge(const Box64& a, const Box64& b) {
    Box64 r;
    for (int k = 0; k < 3; ++k) { r.lo[k] = fmin(a.lo[k], b.lo[k]); r.hi[k] = fmax(a.hi[k], b.hi[k]); }
    return r;
}

// bottom-up fit: the second thread to arrive at a node owns it (its sibling's box is visible behind the fence + atomic).
// Also counts, per leaf, the internal ancestors that SURVIVE the collapse (range longer than leaf_max): the tree's depth.
__global__ void k_fit(const Box64* tri_box, const uint32_t* vals, int n, const uint32_t* child, const uint32_t* parent_node, const uint32_t* parent_leaf,
                      const uint32_t* range, uint32_t leaf_max, Box64* node_box, uint32_t* arrived, uint32_t* max_depth) {
    const int i = (int)(blockIdx.x * BT + threadIdx.x);
    if (i >= n) return;
    uint32_t depth = 0;
    for (uint32_t p = parent_leaf[i];;) {   // depth of this leaf's collapsed leaf = kept ancestors
        if (range[2 * p + 1] - range[2 * p] + 1 > leaf_max) ++depth;
        if (p == 0) break;
        p = parent_node[p];
    }
    atomicMax(max_depth, depth);
    uint32_t node = parent_leaf[i];
    for (;;) {
        __threadfence();
        if (atomicAdd(&arrived[node], 1u) == 0u) return;       // first arrival: the sibling subtree is not done yet
        __threadfence();
        const uint32_t l = child[2 * node], r = child[2 * node + 1];
        const Box64 bl = (l & LEAF_BIT) ? tri_box[vals[l & ~LEAF_BIT]] : node_box[l];
        const Box64 br = (r & LEAF_BIT) ? tri_box[vals[r & ~LEAF_BIT]] : node_box[r];
        node_box[node] = merge(bl, br);
        if (node == 0) return;
        node = parent_node[node];
    }
}

__global__ void k_flags(const uint32_t* range, int n, uint32_t leaf_max, uint32_t* keep) {
    const int i = (int)(blockIdx.x * BT + threadIdx.x);
    if (i < n - 1) keep[i] = (range[2 * i + 1] - range[2 * i] + 1 > leaf_max) ? 1u : 0u;
}

// conservative f32 box of a padded f64 box: Builder::store_box of pt_scene.cpp
__device__ inline void store_box(const Box64& b, float* lo, float* hi) {
    double m = 1e-3;
    for (int k = 0; k < 3; ++k) m = fmax(m, fmax(fabs(b.lo[k]), fabs(b.hi[k])));
    const double pad = 1e-7 * m;
    for (int k = 0; k < 3; ++k) {
        lo[k] = __double2float_rd(b.lo[k] - pad);
        hi[k] = __double2float_ru(b.hi[k] + pad);
    }
}

// kept internal nodes -> BvhNode records; a child whose range has <= leaf_max triangles becomes a triangle leaf
__global__ void k_emit(const uint32_t* child, const uint32_t* range, const uint32_t* keep, const uint32_t* slot, const Box64* node_box,
                       const Box64* tri_box, const uint32_t* vals, int n, BvhNode* out) {
    const int i = (int)(blockIdx.x * BT + threadIdx.x);
    if (i >= n - 1 || !keep[i]) return;
    BvhNode nd;
    uint32_t ref[2];
    for (int c = 0; c < 2; ++c) {
        const uint32_t ch = child[2 * i + c];
        Box64 b;
        if (ch & LEAF_BIT) {
            const uint32_t pos = ch & ~LEAF_BIT;
            b = tri_box[vals[pos]];
            ref[c] = REF_TRIS | pos;                                              // one triangle (count - 1 = 0)
        } else {
            b = node_box[ch];
            if (keep[ch]) ref[c] = REF_NODE | slot[ch];
            else ref[c] = REF_TRIS | ((range[2 * ch + 1] - range[2 * ch]) << 27) | range[2 * ch];
        }
        store_box(b, c == 0 ? nd.lo0 : nd.lo1, c == 0 ? nd.hi0 : nd.hi1);
    }
    nd.child0 = ref[0];
    nd.child1 = ref[1];
    nd.pad0 = nd.pad1 = 0;
    out[slot[i]] = nd;
}

template <class T>
struct DevBuf {
    T* p = nullptr;
    bool alloc(size_t n) { return hipMalloc((void**)&p, std::max<size_t>(n, 1) * sizeof(T)) == hipSuccess; }
    ~DevBuf() { if (p) (void)hipFree(p); }
};

}  // namespace

bool build_blas_device(const TriD* host_tris, uint32_t n, const double mesh_lo[3], const double mesh_hi[3], uint32_t leaf_max, uint32_t max_depth,
                       DeviceBlas& out, hipStream_t st) {
    out.nodes.clear();
    out.order.clear();
    out.depth = 0;
    if (n <= leaf_max || n < 2 || n > 0x07FFFFFFu) return false;   // a single leaf: nothing to build
    DevBuf<TriD> tris;
    DevBuf<Box64> tri_box, node_box;
    DevBuf<uint64_t> keys, keys_sorted;
    DevBuf<uint32_t> vals, vals_sorted, child, parent_node, parent_leaf, range, arrived, keep, slot, depth;
    DevBuf<BvhNode> nodes;
    DevBuf<char> tmp;
    if (!tris.alloc(n) || !tri_box.alloc(n) || !node_box.alloc(n) || !keys.alloc(n) || !keys_sorted.alloc(n) || !vals.alloc(n) || !vals_sorted.alloc(n) ||
        !child.alloc(2 * (size_t)n) || !parent_node.alloc(n) || !parent_leaf.alloc(n) || !range.alloc(2 * (size_t)n) || !arrived.alloc(n) || !keep.alloc(n) ||
        !slot.alloc(n) || !depth.alloc(1) || !nodes.alloc(n))
        return false;
    Box64 mesh;
    for (int k = 0; k < 3; ++k) { mesh.lo[k] = mesh_lo[k]; mesh.hi[k] = mesh_hi[k]; }
    const dim3 grid((n + BT - 1) / BT), block(BT);
    bool ok = hipMemcpyAsync(tris.p, host_tris, (size_t)n * sizeof(TriD), hipMemcpyHostToDevice, st) == hipSuccess;
    ok = ok && hipMemsetAsync(arrived.p, 0, (size_t)n * sizeof(uint32_t), st) == hipSuccess && hipMemsetAsync(depth.p, 0, sizeof(uint32_t), st) == hipSuccess;
    if (!ok) return false;
    hipLaunchKernelGGL(k_keys, grid, block, 0, st, tris.p, n, mesh, tri_box.p, keys.p, vals.p);
    size_t tmp_bytes = 0, scan_bytes = 0;
    if (hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, keys.p, keys_sorted.p, vals.p, vals_sorted.p, (int)n, 0, 63, st) != hipSuccess) return false;
    if (hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, keep.p, slot.p, (int)n - 1, st) != hipSuccess) return false;
    if (!tmp.alloc(std::max(tmp_bytes, scan_bytes))) return false;
    if (hipcub::DeviceRadixSort::SortPairs(tmp.p, tmp_bytes, keys.p, keys_sorted.p, vals.p, vals_sorted.p, (int)n, 0, 63, st) != hipSuccess) return false;
    hipLaunchKernelGGL(k_hierarchy, grid, block, 0, st, keys_sorted.p, (int)n, child.p, parent_node.p, parent_leaf.p, range.p);
    hipLaunchKernelGGL(k_fit, grid, block, 0, st, tri_box.p, vals_sorted.p, (int)n, child.p, parent_node.p, parent_leaf.p, range.p, leaf_max, node_box.p,
                       arrived.p, depth.p);
    hipLaunchKernelGGL(k_flags, grid, block, 0, st, range.p, (int)n, leaf_max, keep.p);
    if (hipcub::DeviceScan::ExclusiveSum(tmp.p, scan_bytes, keep.p, slot.p, (int)n - 1, st) != hipSuccess) return false;
    hipLaunchKernelGGL(k_emit, grid, block, 0, st, child.p, range.p, keep.p, slot.p, node_box.p, tri_box.p, vals_sorted.p, (int)n, nodes.p);
    uint32_t h_depth = 0, last_keep = 0, last_slot = 0;
    ok = hipMemcpyAsync(&h_depth, depth.p, sizeof h_depth, hipMemcpyDeviceToHost, st) == hipSuccess &&
         hipMemcpyAsync(&last_keep, keep.p + (n - 2), sizeof last_keep, hipMemcpyDeviceToHost, st) == hipSuccess &&
         hipMemcpyAsync(&last_slot, slot.p + (n - 2), sizeof last_slot, hipMemcpyDeviceToHost, st) == hipSuccess &&
         hipStreamSynchronize(st) == hipSuccess && hipGetLastError() == hipSuccess;
    if (!ok) return false;
    const uint32_t n_nodes = last_slot + last_keep;
    out.depth = (int)h_depth;
    if (n_nodes == 0 || h_depth > max_depth) return false;        // too deep for the traversal stacks: the caller builds on the host
    out.nodes.resize(n_nodes);
    out.order.resize(n);
    ok = hipMemcpy(out.nodes.data(), nodes.p, (size_t)n_nodes * sizeof(BvhNode), hipMemcpyDeviceToHost) == hipSuccess &&
         hipMemcpy(out.order.data(), vals_sorted.p, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost) == hipSuccess;
    return ok;
}

}  // namespace pt
