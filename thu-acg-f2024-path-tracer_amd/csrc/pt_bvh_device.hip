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
// depth a tree may have, and the builder keeps every tree inside the bound it is given (it no longer falls back).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>

#include "pt_bvh_device.h"

namespace pt {
namespace {

constexpr int BT = 256;


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

// ---- [r3] depth-bounded top-down build over the sorted codes -------------------------------------------------------------------
// Round 2 built Karras' radix tree (one thread per internal node) — whose depth is whatever the codes make it: 30-45 levels for a
// million triangles, far beyond the traversal kernels' LDS stacks (16-24 entries), so every mesh the builder exists for fell back
// to the host. The same tree can be grown top-down, one LEVEL per launch: a node covering the sorted range [first, last] splits
// where the highest differing bit of its first and last code flips (exactly the radix tree's split; equal codes: the middle), and
// the split position is kept inside the window in which neither side exceeds what the levels still available below it can hold
// (leaf_max << levels). Where the bound does not bind the topology IS the radix tree's; where it binds the cut moves to the
// coarsest Morton-cell boundary inside the window (k_split_level) — a subtree is always a contiguous run of the sorted triangles. Nodes are
// numbered in creation order, so every level is a contiguous id range and the boxes are fitted by walking the levels back up.
__device__ inline Box64 merge(const Box64& a, const Box64& b) {
    Box64 r;
    for (int k = 0; k < 3; ++k) { r.lo[k] = fmin(a.lo[k], b.lo[k]); r.hi[k] = fmax(a.hi[k], b.hi[k]); }
    return r;
}
// one level: every node of [level_first, level_end) picks its split and creates its children (leaves are only references)
__global__ void k_split_level(const uint64_t* keys, uint32_t level_first, uint32_t level_end, uint32_t leaf_max, unsigned long long child_cap,
                              uint32_t median_below, uint32_t* node_first, uint32_t* node_last, uint32_t* child, uint32_t* n_nodes) {
    const uint32_t id = level_first + blockIdx.x * BT + threadIdx.x;
    if (id >= level_end) return;
    const uint32_t f = node_first[id], l = node_last[id];
    const unsigned long long n = (unsigned long long)l - f + 1ull;              // > leaf_max: only internal nodes are created
    // the radix-tree split of a run [a, b] of sorted codes: the last position whose code still agrees with keys[a] in the highest bit
    // in which keys[a] and keys[b] differ (equal codes, or a run the caller wants halved: the middle)
    auto radix_split = [&](uint32_t a, uint32_t b, bool halve) -> uint32_t {
        const uint64_t ka = keys[a], kb = keys[b];
        if (ka == kb || halve) return a + (b - a - 1u) / 2u;
        const int prefix = __clzll((long long)(ka ^ kb));
        uint32_t lo = a, hi = b;                                                 // keys[lo] shares more than `prefix` bits with ka, keys[hi] does not
        while (hi - lo > 1u) {
            const uint32_t mid = lo + (hi - lo) / 2u;
            const uint64_t x = ka ^ keys[mid];
            if (x == 0ull || __clzll((long long)x) > prefix) lo = mid; else hi = mid;
        }
        return lo;
    };
    uint32_t s = radix_split(f, l, n <= (unsigned long long)median_below);       // last position of the left child
    // depth bound: both sides must fit into the levels below (child_cap triangles each); n <= 2 child_cap holds by induction.
    // When the radix split violates it, the cut moves to the COARSEST Morton-cell boundary inside the admissible window of
    // positions — the radix split of the window's own run — instead of to the window's edge: a cut through the middle of a cell
    // gives two overlapping boxes (measured: clamping to the edge, or halving runs, costs 25-70 % of K2 on a 1.3 M-triangle mesh).
    const unsigned long long nl = (unsigned long long)s - f + 1ull;
    if (nl > child_cap || n - nl > child_cap) {
        const uint32_t s_min = n > child_cap ? f + (uint32_t)(n - child_cap) - 1u : f;          // smallest admissible last-left position
        const uint32_t s_max = child_cap < n - 1ull ? f + (uint32_t)child_cap - 1u : l - 1u;   // largest
        s = s_min >= s_max ? s_min : radix_split(s_min, s_max + 1u, false);
        if (s > s_max) s = s_max;
    }
    const uint32_t cf[2] = {f, s + 1u}, cl[2] = {s, l};
    for (int c = 0; c < 2; ++c) {
        const uint32_t cnt = cl[c] - cf[c] + 1u;
        if (cnt <= leaf_max) {
            child[2 * id + c] = REF_TRIS | ((cnt - 1u) << 27) | cf[c];
        } else {
            const uint32_t nid = atomicAdd(n_nodes, 1u);
            node_first[nid] = cf[c];
            node_last[nid] = cl[c];
            child[2 * id + c] = REF_NODE | nid;
        }
    }
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
// boxes of one level (the deeper levels are done): child boxes into the BvhNode record, their union kept for the parent
__global__ void k_fit_level(const uint32_t* child, uint32_t level_first, uint32_t level_end, const Box64* tri_box, const uint32_t* vals, Box64* node_box,
                            BvhNode* out) {
    const uint32_t id = level_first + blockIdx.x * BT + threadIdx.x;
    if (id >= level_end) return;
    BvhNode nd;
    Box64 bc[2];
    for (int c = 0; c < 2; ++c) {
        const uint32_t ref = child[2 * id + c];
        if ((ref & REF_TYPE_MASK) == REF_TRIS) {
            const uint32_t first = ref & 0x07FFFFFFu, count = ((ref >> 27) & 7u) + 1u;
            bc[c] = tri_box[vals[first]];
            for (uint32_t k = 1; k < count; ++k) bc[c] = merge(bc[c], tri_box[vals[first + k]]);
        } else {
            bc[c] = node_box[ref & 0x3FFFFFFFu];
        }
        store_box(bc[c], c == 0 ? nd.lo0 : nd.lo1, c == 0 ? nd.hi0 : nd.hi1);
    }
    node_box[id] = merge(bc[0], bc[1]);
    nd.child0 = child[2 * id];
    nd.child1 = child[2 * id + 1];
    nd.pad0 = nd.pad1 = 0;
    out[id] = nd;
}

template <class T>
struct DevBuf {
    T* p = nullptr;
    bool alloc(size_t n) { return hipMalloc((void**)&p, std::max<size_t>(n, 1) * sizeof(T)) == hipSuccess; }
    ~DevBuf() { if (p) (void)hipFree(p); }
};

}  // namespace

bool build_blas_device(const TriD* host_tris, uint32_t n, const double mesh_lo[3], const double mesh_hi[3], uint32_t leaf_max, uint32_t max_depth,
                       uint32_t median_below, DeviceBlas& out, hipStream_t st) {
    out.nodes.clear();
    out.order.clear();
    out.depth = 0;
    if (n <= leaf_max || n < 2 || n > 0x07FFFFFFu || leaf_max < 1 || leaf_max > 8 || max_depth < 1 || max_depth > 40) return false;   // a single leaf: nothing to build
    if (((unsigned long long)leaf_max << max_depth) < (unsigned long long)n) return false;      // cannot fit into max_depth levels at all
    DevBuf<TriD> tris;
    DevBuf<Box64> tri_box, node_box;
    DevBuf<uint64_t> keys, keys_sorted;
    DevBuf<uint32_t> vals, vals_sorted, child, node_first, node_last, counter;
    DevBuf<BvhNode> nodes;
    DevBuf<char> tmp;
    if (!tris.alloc(n) || !tri_box.alloc(n) || !node_box.alloc(n) || !keys.alloc(n) || !keys_sorted.alloc(n) || !vals.alloc(n) || !vals_sorted.alloc(n) ||
        !child.alloc(2 * (size_t)n) || !node_first.alloc(n) || !node_last.alloc(n) || !counter.alloc(1) || !nodes.alloc(n))
        return false;
    Box64 mesh;
    for (int k = 0; k < 3; ++k) { mesh.lo[k] = mesh_lo[k]; mesh.hi[k] = mesh_hi[k]; }
    const dim3 grid((n + BT - 1) / BT), block(BT);
    if (hipMemcpyAsync(tris.p, host_tris, (size_t)n * sizeof(TriD), hipMemcpyHostToDevice, st) != hipSuccess) return false;
    hipLaunchKernelGGL(k_keys, grid, block, 0, st, tris.p, n, mesh, tri_box.p, keys.p, vals.p);
    size_t tmp_bytes = 0;
    if (hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, keys.p, keys_sorted.p, vals.p, vals_sorted.p, (int)n, 0, 63, st) != hipSuccess) return false;
    if (!tmp.alloc(tmp_bytes)) return false;
    if (hipcub::DeviceRadixSort::SortPairs(tmp.p, tmp_bytes, keys.p, keys_sorted.p, vals.p, vals_sorted.p, (int)n, 0, 63, st) != hipSuccess) return false;
    // root = node 0 over the whole sorted range; then one launch per level (a host round trip each: <= max_depth of them)
    const uint32_t root_range[2] = {0u, n - 1u}, one = 1u;
    bool ok = hipMemcpyAsync(node_first.p, &root_range[0], sizeof(uint32_t), hipMemcpyHostToDevice, st) == hipSuccess &&
              hipMemcpyAsync(node_last.p, &root_range[1], sizeof(uint32_t), hipMemcpyHostToDevice, st) == hipSuccess &&
              hipMemcpyAsync(counter.p, &one, sizeof(uint32_t), hipMemcpyHostToDevice, st) == hipSuccess;
    if (!ok) return false;
    std::vector<uint32_t> level_start{0u};          // level L = node ids [level_start[L], level_start[L + 1])
    uint32_t created = 1;
    for (uint32_t level = 0; level_start.back() < created; ++level) {
        if (level >= max_depth) return false;                                   // (cannot happen: the clamp keeps every subtree inside its budget)
        const uint32_t first = level_start.back(), end = created;
        level_start.push_back(end);
        const unsigned long long child_cap = (unsigned long long)leaf_max << (max_depth - level - 1u);   // what a child's subtree may hold
        hipLaunchKernelGGL(k_split_level, dim3((end - first + BT - 1) / BT), block, 0, st, keys_sorted.p, first, end, leaf_max, child_cap, median_below, node_first.p,
                           node_last.p, child.p, counter.p);
        if (hipMemcpyAsync(&created, counter.p, sizeof created, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return false;
        if (created > n) return false;
    }
    const uint32_t n_levels = (uint32_t)level_start.size() - 1u;               // level_start = {0, .., created}
    for (uint32_t level = n_levels; level-- > 0;) {
        const uint32_t first = level_start[level], end = level_start[level + 1];
        hipLaunchKernelGGL(k_fit_level, dim3((end - first + BT - 1) / BT), block, 0, st, child.p, first, end, tri_box.p, vals_sorted.p, node_box.p, nodes.p);
    }
    if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) return false;
    out.depth = (int)n_levels;          // depth of the deepest leaf (root = 0), counted like the host builder's depth_reached
    out.nodes.resize(created);
    out.order.resize(n);
    ok = hipMemcpy(out.nodes.data(), nodes.p, (size_t)created * sizeof(BvhNode), hipMemcpyDeviceToHost) == hipSuccess &&
         hipMemcpy(out.order.data(), vals_sorted.p, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost) == hipSuccess;
    return ok;
}

}  // namespace pt
