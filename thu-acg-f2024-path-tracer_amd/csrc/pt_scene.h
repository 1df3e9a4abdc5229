// Host scene graph (the builder side of World / Hittable / BxDFMaterial / Texture) and its
// flattening into the device tables of pt_types.h.
#pragma once
#include <hip/hip_runtime_api.h>

#include <map>
#include <string>
#include <vector>

#include "pt_host_math.h"
#include "pt_types.h"

struct pt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    int n_cus = 0;
    std::string name;
};

namespace pt {

int set_error(const std::string& msg);   // returns -1
const char* last_error();
bool hip_ok(hipError_t e, const char* what);
// Experiment switches (PT_POOL_SLOTS, PT_SHADE_VARIANT, ...) are read only when PT_EXPERIMENT=1 is set: a release
// library's behaviour does not depend on stray environment variables, and an explicit option always wins over them.
const char* exp_env(const char* name);

enum ObjKind { OBJ_SPHERE, OBJ_QUAD, OBJ_CUBOID, OBJ_MESH, OBJ_INSTANCE };

struct HostTex {
    TexD d{};
    std::vector<uint8_t> image;   // TEX_IMAGE payload (RGB8)
    std::vector<float> image_f;   // TEX_IMAGE_F32 payload (RGB f32)
    bool is_rgb = false;
};
struct HostObj {
    ObjKind kind;
    int mat = -1;
    SphereD sphere{};
    std::vector<QuadD> quads;       // 1 (quad) or 6 (cuboid)
    std::vector<TriD> tris;         // mesh
    std::vector<TriAttr> tri_attr;  // mesh with normals and/or uvs
    bool has_normals = false, has_uvs = false;
    int child = -1;                 // instance
    InstD xf{};
    // An object may be placed in the world directly any number of times and wrapped by any number of instances, and an instance
    // may wrap an instance (Instance::new / World::add_object take an Arc<dyn Hittable>, instance.rs:20-30, world.rs:18-24).
};

struct DeviceBuffers {
    std::vector<void*> allocs;
    SceneD view{};
    void release();
};

}  // namespace pt

struct pt_scene {
    pt_ctx* ctx = nullptr;
    std::vector<pt::HostTex> tex;
    std::vector<pt::MatD> mats;
    std::vector<pt::HostObj> objs;
    std::vector<int> world_objects, world_lights;
    std::map<std::string, int> images;   // registered image name -> texture handle
    bool built = false;
    bool float_hdr = false;        // scene scripts / host mirrors load Radiance files as f32 textures (pt_scene_set_float_hdr)
    uint32_t n_prims = 0;
    uint32_t n_mesh_entries = 0;   // world-level triangle meshes (picks the K2 variant)
    bool motionless = false;       // no sphere moves (p1 == p2 everywhere): a ray's time never reaches an arithmetic result
    uint32_t stack_need = 0;       // worst-case traversal stack entries of this scene's BVHs
    uint32_t device_bvh_min_tris = 1u << 19;   // meshes with at least this many triangles get their BVH built on the GPU (0 = never)
    uint32_t n_device_blas = 0, device_blas_depth = 0;   // meshes of the last build that the GPU builder handled, deepest of them
    uint32_t stack_need_extend2 = 0;   // ... for k_extend2, whose top-level walk is stackless when the entry list is walked flat
    pt::DeviceBuffers dev;
    // path pool cache (re-used across pt_render calls of the same size)
    void* pool_mem = nullptr;
    size_t pool_bytes = 0;
    double* tile_accum = nullptr;  // dynamic mode: the frame accumulator in tile order (PoolD::accum_tiled), re-used like the pool
    size_t tile_accum_bytes = 0;
    uint32_t* compact_scratch = nullptr;   // the end-of-frame compaction's hole / mover lists + counters (pt_render.cpp)
    size_t compact_scratch_words = 0;
    pt::CountersD* d_counters = nullptr;
    pt::CountersD* h_counters = nullptr;   // pinned
    ~pt_scene();
};

namespace pt {
int scene_build(pt_scene* s);   // flatten + BVH + upload
}
