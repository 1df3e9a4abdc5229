// Flattened scene + path-pool layout shared by the host scene builder and the HIP kernels.
// Everything the kernels read lives in HBM as plain arrays of these PODs (records for the per-path
// state and for the small read-only scene tables that sit in L2).
#pragma once
#include <stdint.h>

namespace pt {

// ---- BVH -----------------------------------------------------------------------------
// One node format for the top-level tree (over world objects) and every per-mesh tree.
// Boxes are stored as f32 rounded OUTWARD from the padded f64 bounds (conservative); the slab
// test runs in f32 with a per-ray error margin (pt_kernels.hip, slab_f32) — box tests only have
// to be conservative, the f64 primitive tests decide the result. 64 B = four 16-B loads per visit.
// Child reference (32 bit):
//   00xx.. internal node index
//   01cc cfff.. triangle leaf: ccc = count-1 (1..8), f = first triangle (BLAS order)
//   10xx.. world entry index (top-level leaf, always exactly one entry)
//   0xFFFFFFFF empty slot, 0xC0000000 "leave instance" sentinel (stack only)
struct alignas(16) BvhNode {
    float lo0[3], hi0[3];
    float lo1[3], hi1[3];
    uint32_t child0, child1;
    uint32_t pad0, pad1;
};
constexpr uint32_t REF_TYPE_MASK = 0xC0000000u;
constexpr uint32_t REF_NODE = 0x00000000u;
constexpr uint32_t REF_TRIS = 0x40000000u;
constexpr uint32_t REF_ENTRY = 0x80000000u;
constexpr uint32_t REF_EMPTY = 0xFFFFFFFFu;
constexpr uint32_t REF_LEAVE_INSTANCE = 0xC0000000u;
constexpr int TRAVERSAL_STACK = 32;     // LDS entries per lane
constexpr int MAX_BLAS_DEPTH = 20;      // enforced by the host builder
constexpr int MAX_TLAS_DEPTH = 10;

// ---- geometry ------------------------------------------------------------------------
enum PrimKind : uint32_t { PRIM_SPHERE = 0, PRIM_QUAD = 1, PRIM_TRI = 2 };
enum EntryKind : uint32_t { ENTRY_SPHERE = 0, ENTRY_QUAD = 1, ENTRY_CUBOID = 2, ENTRY_MESH = 3 };
constexpr uint32_t PRIM_HAS_NORMALS = 1u << 8, PRIM_HAS_UVS = 1u << 9;
constexpr uint32_t PRIM_MAT_KIND_SHIFT = 16;   // PrimRef::kind bits 16..23: MatKind of the primitive's material (k_shade's class sort)

struct PrimRef {         // indexed by GLOBAL primitive id (lights list first, then objects)
    uint32_t kind;       // PrimKind | PRIM_HAS_*
    uint32_t index;      // index into spheres[] / quads[] / tris[]
    uint32_t mat;
    int32_t inst;        // OUTERMOST instance of the placement's chain, or -1
};
struct Entry {           // one per world-level object (lights list first, then objects)
    uint32_t kind;       // EntryKind
    uint32_t first_prim; // global id of its first primitive
    int32_t inst;        // OUTERMOST instance of the placement's chain, or -1
    uint32_t blas_root;  // node index (ENTRY_MESH); several placements of one mesh share the tree and its triangles
    float extent;        // ENTRY_MESH: max |coordinate| of the mesh's (local-space) BVH boxes
    uint32_t n_prims;    // 1 (sphere, quad), 6 (cuboid) or the triangle count (mesh)
    uint32_t pad[2];
};
// Flat top-level walk (SceneD::tlas_flat): everything a step of the walk needs in ONE 64-byte record, i.e. one scalar load — the
// entry's world-space box (padded and rounded outward like the node boxes), a copy of its Entry, and for spheres / quads /
// cuboids the record index of the (first) primitive, so that neither Entry nor PrimRef has to be fetched first: with a
// handful of waves per SIMD every dependent scalar load is ~200 exposed cycles, and the walk made three per entry.
struct EntryBox {
    float lo[3], hi[3];
    uint32_t entry;      // index into SceneD::entries
    uint32_t kind;       // EntryKind                                    -- Entry's fields from here
    uint32_t first_prim;
    int32_t inst;
    uint32_t blas_root;
    float extent;
    uint32_t n_prims;
    uint32_t prim_kind;  // PRIM_SPHERE / PRIM_QUAD of the entry's primitives, or ENTRYBOX_VIA_PRIMREF: look them up in prims[]
    uint32_t prim_index; // spheres[] / quads[] index of primitive first_prim (the faces of a cuboid follow consecutively)
    uint32_t pad;
};
constexpr uint32_t ENTRYBOX_VIA_PRIMREF = 0xFFFFFFFFu;
// Parallel to SceneD::entry_box, meaningful for ENTRY_CUBOID records: the cuboid's OBJECT-space box (cuboid.rs:11-58 builds its six
// quads on the faces of [min, max]), f32 rounded outward like every stored box; EntryBox::extent then holds max |coordinate| of
// it. The flat walk slab-tests the object-space ray against it and only runs the exact quad tests of the faces the ray can
// enter or leave through (flat_top_level, "face culling"): typically two of the six.
struct CuboidBox {
    float lo[3], hi[3];
    float pad[2];
};
struct SphereD { double r, p1[3], p2[3]; };
struct QuadD { double q[3], u[3], v[3], w[3], n[3], d; };
struct TriD { double v0[3], v1[3], v2[3]; };
struct TriAttr { double n[3][3]; double uv[3][2]; };   // only when the mesh has them
// One Instance of a placement (instance.rs:12-30): forward transform (columns c0..c2, translation t) and its analytic rigid
// inverse. Instances nest (Instance::new takes any Hittable): a placement is a CHAIN of them — `inner` is the next one
// towards the object, `outer` the next one towards the world, -1 at the ends. Every placement owns its chain.
struct InstD {
    double c0[3], c1[3], c2[3], t[3], i0[3], i1[3], i2[3], it[3];
    int32_t inner, outer;
};

// ---- textures / materials ------------------------------------------------------------
// TEX_IMAGE_F32: ImageTexture WITHOUT the `.to_rgb8()` squash of texture.rs:67 — the decoder's f32 samples (a Radiance .hdr decodes to
// Rgb32F) live in SceneD::atlas_f and the lookup of texture.rs:73-91 returns them widened to f64 (behind a flag).
enum TexKind : uint32_t { TEX_SOLID_RGB = 0, TEX_SOLID_F = 1, TEX_CHECKER = 2, TEX_IMAGE = 3, TEX_IMAGE_F32 = 4 };
struct TexD {
    uint32_t kind, t1, t2, w, h;
    uint32_t flat;       // TEX_CHECKER whose two children are solid: their values sit in c1 / c2 — one level less in the chain of
                         // DEPENDENT loads primitive -> material -> texture -> child texture (~700 cycles each in k_shade).
                         // (Also copying the descriptors into the material record made k_shade spill 384 B per lane.)
    uint64_t ofs;        // byte offset into the RGB8 atlas (TEX_IMAGE) / element offset into the f32 atlas (TEX_IMAGE_F32)
    double v[3];
    double inv_scale;
    double c1[3], c2[3];
};
enum MatKind : uint32_t { MAT_DIFFUSE = 0, MAT_METAL = 1, MAT_GLASS = 2, MAT_PRINCIPLED = 3, MAT_LIGHT = 4,
                          MAT_SHEEN = 5, MAT_CLEARCOAT = 6, MAT_MIX = 7, MAT_KINDS = 8 };
constexpr uint32_t CLASS_MISS = 0u, CLASS_IDLE = 1u + MAT_KINDS, CLASS_DEAD = 2u + MAT_KINDS, N_CLASSES = 3u + MAT_KINDS;
// MAT_SHEEN: p[0..2] = base colour, p[3] = sheen_tint (sheen.rs). MAT_CLEARCOAT: alpha_g (clearcoat.rs).
// MAT_MIX: p[0] = t, color_tex / rough_tex hold the two child MATERIAL indices (mix.rs; children are leaves).
struct MatD {
    uint32_t kind;
    int32_t color_tex, rough_tex, nmap_tex;
    double ior;
    // principled.rs:45-58 order: metallic, roughness, subsurface, specular, specular_tint,
    // ior, spec_trans, sheen, sheen_tint, clearcoat, clearcoat_gloss
    double p[11];
    double lobe_w[4], lobe_p[4], alpha_g;   // principled.rs:75-100, precomputed on the host
    // A SOLID colour / roughness texture's value, copied here by the scene build: the material record then answers without the
    // texture descriptor — one level less in k_shade's chain of dependent loads (primitive -> material -> texture). Filled for
    // MAT_DIFFUSE / METAL / GLASS / PRINCIPLED / LIGHT; a mix's children are looked up the long way.
    uint32_t color_solid, rough_solid;
    double color_v[3], rough_v;
};

// ---- camera --------------------------------------------------------------------------
struct CamD {
    double center[3], pixel00[3], pixel_du[3], pixel_dv[3], dof_right[3], dof_up[3];
    double blur_strength;
    double env_color[3];
    double two_pi_scale;     // rand's UniformFloat scale for gen_range(0.0..=2pi)
    uint32_t width, height, max_depth, env_is_map;
    int32_t env_tex;
    uint32_t n_lights;
    uint32_t lens_zero;      // defocus radius 0 (dof_right = dof_up = 0, no -0.0 in center): the lens point is `center` exactly
    uint32_t motionless;     // no moving sphere in the scene: Ray::time is drawn (camera.rs:165) but its value is never used
};

struct SceneD {
    const BvhNode* nodes;
    const Entry* entries;
    const PrimRef* prims;
    const SphereD* spheres;
    const QuadD* quads;
    const TriD* tris;
    const TriAttr* tri_attr;     // parallel to tris when any mesh has normals/uvs, else null
    const uint32_t* tri_gid;     // BLAS-order triangle -> face index inside its mesh; global id = Entry::first_prim + that (a mesh may be placed several times)
    const InstD* insts;
    const TexD* tex;
    const MatD* mats;
    const uint8_t* atlas;
    const float* atlas_f;        // f32 RGB texels of the TEX_IMAGE_F32 textures
    const uint32_t* lights;      // entry indices of the lights list
    uint32_t tlas_root;          // child reference of the top-level root
    float tlas_extent;           // max |coordinate| of the top-level BVH boxes
    uint32_t n_entries, n_prims, n_lights;
    const CuboidBox* cuboid_box; // parallel to entry_box (cuboid records only)
    const EntryBox* entry_box;   // the flat top level's walk list: one record per entry, NON-MESH entries first (each group in entry order)
    uint32_t tlas_flat;          // n_entries <= TLAS_FLAT_MAX: K2 walks the entry list instead of the top-level tree
    uint32_t flat_pairs;         // tlas_flat and the scene has cuboids: the batch K2 runs its (ray, primitive) pair passes
};
constexpr uint32_t TLAS_FLAT_MAX = 24;   // round 1 (vector loads): 8-10 entries -26 % / -7 % K2 time, 17 entries (scene 5) +20 % -> limit 12;
                                          // round 2 (scalar loads, ldu): 17 entries -20 % -> limit raised

// ---- path pool (one slot per resident path) --------------------------------------------
// Two work-assignment modes:
//  static  (slots_per_pixel = k >= 1): slot s owns pixel (s % n_pixels) and renders samples
//          spp_begin + (s / n_pixels) + j*k into its own accumulator ax/ay/az; deterministic, and
//          for k = 1 exactly the reference's per-pixel sample order (camera.rs:106-108).
//  dynamic (default): a finished path pulls the next (pixel, sample) work item from one global
//          counter — wave-aggregated: ballot + popcount + one atomic per wave — so every lane
//          stays busy until the frame's sample budget is exhausted; radiance is added to the
//          frame accumulator with hardware f64 atomics. Work items are ordered sample-major and,
//          inside a sample, by 8x8 PIXEL TILES (item % (tiles*64) -> tile, pixel in tile), so that
//          the 64 lanes of a wave start on one compact tile: primary rays — half of all segments
//          in scene 6 — stay coherent in traversal and in material. Items that fall outside a
//          ragged image edge are skipped (slot state SLOT_IDLE for one iteration).
constexpr uint32_t HIT_NONE = 0xFFFFFFFFu;
// PoolD::hit_prim — what K2 hands to K3, one word per slot: the closest primitive's global id in bits 0..27 (all ones:
// nothing hit) and in bits 28..31 the CLASS k_shade sorts the slot by, so that K3's classification is ONE coalesced
// load per slot instead of a chain of three dependent ones (state -> id -> PrimRef): 0 miss, 1 + MatKind of the hit
// primitive's material, then idle and dead slots (K2 reads the slot state anyway).
constexpr uint32_t HIT_ID_MASK = 0x0FFFFFFFu, HIT_CLASS_SHIFT = 28;
constexpr uint32_t HIT_SLOT_IDLE = 0xFFFFFFFEu, HIT_SLOT_DEAD = 0xFFFFFFFDu;   // K2-internal sentinels next to HIT_NONE
constexpr uint32_t SLOT_DEAD = 0xFFFFFFFFu;   // value of `bounce` for a finished slot
constexpr uint32_t SLOT_IDLE = 0xFFFFFFFEu;   // dynamic mode: drew an item outside the image, draws again next iteration
// Path state is kept as two records per slot (array of structures): k_shade visits slots in
// material-class order and k_extend2's mesh pass visits them compacted, i.e. both PERMUTED inside a
// window — with one array per field every wave touched 1/8 of many 128-B lines and the rest of each
// line had left the L2 before the wave that needed it came by (measured 2.6 GB of HBM traffic per
// k_shade launch for 1.1 GB of state). A record is read/written whole by its own lane (16 B at a time).
// K2 reads RayRec only; K3 reads and writes both: 96 B in, 96 B out per segment.
struct alignas(64) RayRec {                       // current ray (direction normalised) + RNG position of the path
    double ox, oy, oz, dx, dy, dz, time;
    uint32_t sample, draw;                        // sample index, RNG draw counter
};
// 32 B. Measured (round 2, -DPT_PATHREC_BYTES=64): padding the record to a whole 64-B sector and writing all of it —
// so that K3 never writes half sectors — made K3 6 % SLOWER (560 vs 525 ms per 1000-spp frame of scene 6): two more 16-B
// stores per slot cost more than the partial-sector writes do. The 64-B form stays behind the macro for that measurement.
#ifndef PT_PATHREC_BYTES
#define PT_PATHREC_BYTES 32
#endif
struct alignas(PT_PATHREC_BYTES) PathRec {
    double tx, ty, tz;                            // throughput
    uint32_t pixel, pad;                          // dynamic mode: pixel of the sample in flight
#if PT_PATHREC_BYTES == 64
    double reserved[4];
#endif
};
struct PoolD {
    RayRec* ray;
    PathRec* path;
    double *ax, *ay, *az;                         // static mode: sum over this slot's finished samples
    double *rx, *ry, *rz;                         // static mode: radiance of the sample in flight (camera.rs:172); the dynamic
                                                  // mode adds every contribution to the frame accumulator right away
    uint32_t* hit_prim;
    uint32_t* bounce;                             // bounce count, or SLOT_DEAD / SLOT_IDLE
    double* accum;                                // dynamic mode: frame accumulator (W*H*3 sums). accum_tiled: three CHANNEL PLANES of
                                                  // n_tile_pixels sums each, a plane in the order the work items are handed out — tile
                                                  // after tile, 64 pixels of an 8x8 tile consecutive (index = plane + tile*64 + (y&7)*8 + (x&7)):
                                                  // the lanes of a group that finish together mostly hold pixels of the same few tiles, so one
                                                  // global_atomic_add_f64 wave-instruction covers runs of consecutive 8-byte words — whole 64-byte
                                                  // memory-side atomic requests (MI355X_MICROARCH.md "Global float atomics": they execute at the
                                                  // memory side, one request per 64 B touched) — instead of 64 requests at a 24-byte stride.
                                                  // k_detile adds the planes to the caller's (y, x, c) accumulator once per render.
    double inv_width;                             // 1 / width (pixel -> row by one multiplication and an exact correction)
    uint32_t accum_tiled;
    unsigned long long total_work;                // dynamic mode: n_pixels * (spp_end - spp_begin)
    uint32_t n_slots, n_pixels, k;                // k = slots per pixel (static mode)
    uint32_t spp_begin, spp_end;
    uint32_t dynamic, n_alloc;                    // n_alloc: slots rounded up to a multiple of 64
    uint32_t defer_regen, compact;                   // dynamic mode: a path that ends ON A SURFACE (roulette, sampler, depth) parks its slot
                                                  // as SLOT_IDLE; it is refilled next iteration among the idle slots (k_shade)
    uint32_t width, height, tiles_x, n_tile_pixels;   // dynamic mode: 8x8 tiling, n_tile_pixels = tiles_x*tiles_y*64
};

// The work counter of the dynamic mode is SHARDED: one word saturates at ~88 dequeues/us on this
// chip (MI355X_MICROARCH.md, row "dequeue") and a frame needs one dequeue per wave per iteration
// (65k per launch at 4M resident paths), which alone would cost ~0.75 ms per launch. Shard s hands
// out the work items w with w % WORK_SHARDS == s; a block always uses shard blockIdx.x % WORK_SHARDS.
constexpr uint32_t WORK_SHARDS = 64;
constexpr int PROF_COLS = 12;   // diagnostic build (-DPT_STAMPS): columns of the per-class profile
struct CountersD {
    unsigned long long alive;        // slots still rendering
    unsigned long long segments;     // extend() calls on live paths
    unsigned long long samples;      // finished samples
    // window queues of k_extend2 / k_shade<sort>: blocks draw the next window of the pool from these instead
    // of striding over it (window costs differ by an order of magnitude between sky and mesh tiles). Each
    // kernel zeroes the OTHER kernel's queue; the two alternate on one stream.
    unsigned long long win_extend, win_shade;
    unsigned long long pad[11];
    // diagnostic builds only (-DPT_STAMPS, tools/build_variant.sh): wave-cycle sums per k_shade class
    // [class][0 groups, 1 record-load wait, 2 body, 3 work dequeue, 4 regeneration + stores, 5 whole], [N_CLASSES][..] = window phases
    unsigned long long prof[N_CLASSES + 1][PROF_COLS];
    struct alignas(128) Shard { unsigned long long next; unsigned long long pad[15]; } work[WORK_SHARDS];
};

}  // namespace pt
