/* pt_amd.h — C ABI of the MI355X-native wavefront path tracer (libpt_amd.so).
 *
 * Drop-in boundary for ONE path of chiefchewie/thu-acg-f2024-path-tracer: the per-pixel
 * integrator entered through `Camera::render(&self, world: &World, filename)`
 * (src/camera.rs:79) and everything below it (camera.rs trace loop, hittable/ BVH traversal
 * and primitive intersection, bsdf/ sample+pdf+eval). The scene / material / camera builder
 * calls mirror the reference's constructors one-to-one so that its main.rs scene scripts map
 * onto this header line by line (INTEGRATION.md shows the Rust `extern "C"` binding).
 *
 * Conventions: plain C types only; every call returns 0 (or a handle >= 0) on success and
 * -1 on error with pt_last_error() describing it; nothing panics or throws across the ABI.
 * Handles (textures, materials, objects) are small ints local to one pt_scene. A pt_ctx and
 * its scenes are not thread-safe: one render at a time per context. There is NO CPU
 * fallback: every entry point that computes needs the HIP device and fails loudly without.
 *
 * Arithmetic: the reference computes in f64 (src/vec3.rs:3-6); so do these kernels.
 */
#ifndef PT_AMD_H
#define PT_AMD_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct pt_ctx pt_ctx;       /* one HIP device + stream */
typedef struct pt_scene pt_scene;   /* World + its textures/materials/objects (hittable/world.rs:5-8) */

/* The 12 public fields of `Camera` (src/camera.rs:23-36). environment: Color(Vec3) or
 * Map(ImageTexture) (camera.rs:16-19) -> env_is_map + env_color / env_tex. */
typedef struct pt_camera {
    double aspect_ratio;
    uint32_t image_width, samples_per_pixel, max_depth, env_is_map;
    double vfov;
    double look_from[3], look_at[3], vup[3];
    double blur_strength, focal_length, defocus_angle;
    double env_color[3];
    int32_t env_tex;
    int32_t _pad;
} pt_camera;

typedef struct pt_render_opts {
    uint32_t slots_per_pixel;   /* resident paths per pixel; 0 = auto (fills the GPU; about one path per 30 samples of the
                                   frame, at most 1 M paths per CU: 134 M paths = 14 GB of device memory for 1920x1080 @
                                   4000 spp on an MI355X, 28 GB at most, kept by the scene until pt_scene_destroy); 1 = the reference's
                                   exact per-pixel sample order */
    uint32_t accum_on_device;   /* accum points to device memory (e.g. a torch tensor) */
    uint32_t profile;           /* time every kernel launch with HIP events */
    uint32_t overwrite;         /* 0: the frame's sums are ADDED to accum (sample ranges accumulate); 1: accum is overwritten */
    void* stream;               /* hipStream_t to launch on; NULL = the context's own stream */
} pt_render_opts;

typedef struct pt_render_stats {
    uint64_t samples, segments, iterations;
    uint32_t n_slots, slots_per_pixel;
    double ms_total;                       /* wall time inside pt_render (host clock, synced) */
    double ms_extend, ms_shade, ms_other;  /* HIP-event time per kernel family (profile=1) */
    uint64_t launches_extend, launches_shade;
    uint32_t extend_variant, shade_variant;   /* K2: 0 two-phase k_extend2, 1 batch k_extend; K3: sort*10 + min waves/SIMD */
    uint32_t blocks_extend, blocks_shade;
    uint32_t compactions;                  /* times the thinning pool was compacted at the frame's end (dynamic mode) */
    uint32_t n_alloc_end;                  /* slots the last launches still covered */
} pt_render_stats;

const char* pt_last_error(void);
int pt_set_error_message(const char* msg);   /* host layers above the ABI report through the same channel; returns -1 */
/* Fails (-1) when no HIP device is present: there is no host fallback. */
int pt_ctx_create(int device, pt_ctx** out);
void pt_ctx_destroy(pt_ctx*);
int pt_device_name(pt_ctx*, char* buf, uint32_t n);

pt_scene* pt_scene_create(pt_ctx*);
void pt_scene_destroy(pt_scene*);
pt_ctx* pt_scene_ctx(pt_scene*);

/* ---- textures: src/texture.rs ---------------------------------------------------------- */
int pt_tex_solid_rgb(pt_scene*, double r, double g, double b);          /* SolidTexture<Vec3> :11-25 */
int pt_tex_solid_f(pt_scene*, double v);                                /* SolidTexture<f64>       */
int pt_tex_checker(pt_scene*, double scale, int tex1, int tex2);        /* CheckerTexture::new :34-40 */
int pt_tex_image_rgb8(pt_scene*, uint32_t w, uint32_t h, const uint8_t* rgb);   /* ImageTexture (decoded, RGB8) :56-70 */
/* The float-HDR option: an ImageTexture that keeps the decoder's f32 samples instead of `.to_rgb8()`
 * (texture.rs:67) — same nearest-texel lookup (texture.rs:73-91), values not clamped to [0,1]. Usable wherever an image texture
 * is (colour, normal map, Camera::environment). pt_scene_set_float_hdr(scene, 1) makes the scene scripts (pt_build_scene) and the
 * host mirrors load Radiance .hdr files this way; the default (0) is the reference's RGB8 behaviour. */
int pt_tex_image_rgbf32(pt_scene*, uint32_t w, uint32_t h, const float* rgb);
int pt_scene_set_float_hdr(pt_scene*, int on);
int pt_scene_float_hdr(pt_scene*);
/* ---- materials: src/bsdf/, src/material.rs ---------------------------------------------- */
int pt_mat_diffuse(pt_scene*, int color_tex, int normal_map_tex);       /* DiffuseBRDF::{new,from_rgb,from_textures} diffuse.rs:21-47; -1 = no map */
int pt_mat_metal(pt_scene*, int color_tex, int rough_tex);              /* MetalBRDF::new metal.rs:23-35 */
int pt_mat_glass(pt_scene*, int color_tex, int rough_tex, double anisotropic, double ior);   /* GlassBSDF::new glass.rs:28-40 */
int pt_mat_principled(pt_scene*, int color_tex, const double params[11]);   /* PrincipledBSDF::new principled.rs:45-73, same argument order */
int pt_mat_light(pt_scene*, int emission_tex);                          /* DiffuseLight::new material.rs:155-164 */
/* the three bsdf/ materials no reference scene instantiates */
int pt_mat_mix(pt_scene*, double t, int mat1, int mat2);                /* MixBxDf::new mix.rs:14-20; a child may itself be a mix of non-mix materials (two levels) */
int pt_mat_sheen(pt_scene*, double r, double g, double b, double sheen_tint);   /* SheenBRDF::new sheen.rs:17-22 */
int pt_mat_clearcoat(pt_scene*, double clearcoat_gloss);                /* ClearcoatBRDF::new clearcoat.rs:14-18 */
/* ---- geometry: src/hittable/ ------------------------------------------------------------ */
int pt_sphere(pt_scene*, double radius, const double p1[3], const double p2[3], int mat);   /* Sphere::new_still/new_moving sphere.rs:22-46 */
int pt_quad(pt_scene*, const double q[3], const double u[3], const double v[3], int mat);   /* Quad::new quad.rs:17-36 */
int pt_cuboid(pt_scene*, const double a[3], const double b[3], int mat);                    /* Cuboid::new cuboid.rs:11-58 */
/* TriangleMesh::from_obj mesh.rs:149-197: f32 positions (and optional normals / texcoords,
 * all indexed by the position index as the reference does) + u32 index triples. */
int pt_mesh(pt_scene*, double scale, uint32_t n_pos, const float* pos, uint32_t n_idx, const uint32_t* idx,
            uint32_t n_nrm, const float* nrm, uint32_t n_uv, const float* uv, int mat);
int pt_instance(pt_scene*, int obj, const double axis[3], double angle, const double translation[3]);   /* Instance::new instance.rs:20-30 */
/* ---- world: src/hittable/world.rs:10-29 -------------------------------------------------- */
int pt_world_add_object(pt_scene*, int obj);
int pt_world_add_light(pt_scene*, int obj);
int pt_world_build(pt_scene*);          /* build_bvh: flatten to SoA, build BVHs, upload to HBM */
uint32_t pt_world_prim_count(pt_scene*);
/* BVH::build (bvh.rs:24-121) for large meshes on the GPU: meshes with at least min_triangles triangles get an LBVH built
 * by HIP kernels at the next pt_world_build (default 2^19; 0 = always the host's binned-SAH builder). Any conservative
 * tree gives the same hits; an LBVH is cheaper to build and costlier to traverse. _info: meshes the GPU builder handled
 * in the last build and the depth of the deepest of them. */
int pt_world_set_device_bvh_threshold(pt_scene*, uint32_t min_triangles);
int pt_world_device_bvh_info(pt_scene*, uint32_t* n_meshes, uint32_t* deepest);

/* ---- asset ingest (host): the roles of tobj::load_obj (main.rs:408) and
 * ImageReader::open().decode().to_rgb8() (texture.rs:62-67) for .obj / Radiance .hdr ------- */
int pt_load_obj(const char* path, float** pos, uint32_t* n_pos, uint32_t** idx, uint32_t* n_idx, float** uv, uint32_t* n_uv);
/* OBJ with vn and separate v/vt/vn index streams, expanded to ONE index per corner (tobj's single_index): the fix of
 * mesh.rs:173-184's position-indexed normals / texcoords. Feed the result to pt_mesh. */
int pt_load_obj_single_index(const char* path, float** pos, uint32_t* n_pos, uint32_t** idx, uint32_t* n_idx, float** nrm, uint32_t* n_nrm,
                             float** uv, uint32_t* n_uv);
int pt_load_hdr_rgb8(const char* path, uint8_t** rgb, uint32_t* w, uint32_t* h);
int pt_load_hdr_rgbf32(const char* path, float** rgb, uint32_t* w, uint32_t* h);   /* the same decode without .to_rgb8(): f32 RGB, free with pt_free */
int pt_load_png_rgb8(const char* path, uint8_t** rgb, uint32_t* w, uint32_t* h);   /* PNG -> RGB8 (alpha dropped like to_rgb8, texture.rs:67) */
/* JPEG -> RGB8: baseline and progressive Huffman JPEG (the reference's earthmap.jpg / envmap.jpg, main.rs:100,365) with libjpeg's
 * reference arithmetic (ISLOW integer IDCT, fixed-point YCbCr->RGB, fancy chroma upsampling): csrc/pt_jpeg.cpp */
int pt_load_jpeg_rgb8(const char* path, uint8_t** rgb, uint32_t* w, uint32_t* h);
void pt_free(void*);
/* images decoded by the caller (any format this library has no decoder for — or pixels a host wants to substitute): hand them
 * over decoded, under the file name the reference's scene opens ("envmap.jpg", "earthmap.jpg", "bricks/color.png", ...); a
 * registered image takes precedence over the file */
int pt_register_image(pt_scene*, const char* name, uint32_t w, uint32_t h, const uint8_t* rgb);
int pt_find_registered_image(pt_scene*, const char* name);   /* texture handle or -1 */
int pt_save_png(const char* path, uint32_t w, uint32_t h, const uint8_t* rgb);   /* imgbuf.save camera.rs:118 */

/* ---- the reference's scene scripts main.rs:14-618 (`-s N`); fills the camera the script sets
 * up. scene_seed replaces the unseeded build-time RNG of scene 1 (main.rs:38-47). ----------- */
int pt_build_scene(pt_scene*, int scene_id, uint32_t width, uint32_t spp, const char* asset_dir, uint64_t scene_seed,
                   pt_camera* out_cam);

/* Camera::init camera.rs:51-77; out6x3 = forward,right,up,pixel00,pixel_du,pixel_dv */
int pt_camera_init(const pt_camera*, double out6x3[18], uint32_t* image_height);

/* Camera::render camera.rs:79-126, the hot path. Adds to accum[(y*W+x)*3+c] the SUM over
 * samples [spp_begin, spp_end) of trace(y, x) — sums, not means, so that sample ranges
 * rendered on different GPUs add up (one reduce) before pt_resolve_u8. */
int pt_render(pt_scene*, const pt_camera*, uint64_t seed, uint32_t spp_begin, uint32_t spp_end, double* accum,
              const pt_render_opts* opts, pt_render_stats* stats);
/* camera.rs:109-114,128-130: mean, sqrt gamma, clamp(0,0.999)*256 as u8. Host buffers. */
int pt_resolve_u8(pt_ctx*, const double* accum, uint32_t n_pixels, uint32_t total_spp, uint8_t* rgb8);

/* ---- multi-GPU: one process per GPU, spp sharding, ONE RCCL reduce over xGMI --------------------------------------
 * The reference is a single process (rayon over pixels, camera.rs:102); samples of a pixel are only summed
 * (camera.rs:106-108), so rank r of N renders the sample range pt_shard_range(spp, r, N) of every pixel and one
 * ncclReduce(sum, f64, root 0) of the W*H*3 sample SUMS lands the frame on rank 0. RCCL is called directly from
 * /opt/rocm on the device accumulator, on the render stream: no torch, no host bounce. */
typedef struct pt_comm pt_comm;       /* one rank of an RCCL communicator, bound to a pt_ctx's device and stream */
void pt_shard_range(uint32_t spp, int rank, int world, uint32_t* lo, uint32_t* hi);   /* contiguous, disjoint, near-equal */
/* Rendezvous of the `world` processes of one launch (rank / world as torchrun's RANK / WORLD_SIZE): rank 0 creates the
 * RCCL unique id and publishes it in the file `id_path` (any path all ranks agree on and no earlier launch used), the
 * others wait for it up to timeout_s (<= 0: 120 s); then ncclCommInitRank. world == 1 needs no file. */
int pt_comm_create(pt_ctx*, int rank, int world, const char* id_path, double timeout_s, pt_comm** out);
void pt_comm_destroy(pt_comm*);
int pt_comm_rank(pt_comm*);
int pt_comm_world(pt_comm*);
int pt_comm_barrier(pt_comm*);        /* all ranks have arrived and every device is idle */
int pt_comm_allreduce_f64(pt_comm*, double* host_values, uint32_t n /* <= 64 */, int op /* 0 sum, 1 max */);
int pt_bootstrap_exchange(const char* path, int rank, void* bytes, uint32_t n, double timeout_s);   /* the file rendezvous itself (host only) */
/* Camera::render on all ranks of the communicator: samples [0, spp_total) split by pt_shard_range, rendered into a
 * device accumulator, reduced onto rank 0 and ADDED there to accum_root (host, W*H*3 sums; ignored on other ranks;
 * opts->overwrite: stored instead of added). stats are this rank's. A failure on any rank is agreed on before the
 * reduce is posted: every rank returns -1 and no rank waits; a failure inside a collective aborts the communicator. */
int pt_render_multi(pt_scene*, const pt_camera*, uint64_t seed, uint32_t spp_total, pt_comm*, double* accum_root,
                    const pt_render_opts* opts, pt_render_stats* stats);

/* ---- parity probes (tests only) ---------------------------------------------------------- */
/* closest hit of n rays {o.xyz, d.xyz, time} against the built world (World::intersect_all,
 * world.rs:47-62, ray_t = [1e-3, inf)); out[15*i] = {hit, t, prim_id, u, v, front_face,
 * point.xyz, geometric_normal.xyz, shading_normal.xyz} */
int pt_intersect(pt_scene*, const double* rays, uint32_t n, double* out);
/* elementwise device arithmetic: which = 0 sqrt(a) 1 a/b 2 a*b+a 3 sin 4 cos 5 acos 6 atan2(a,b)
 * 7 pow(a,b) 8 log2 9 rng uniform(seed=a, pixel=b, sample=7, draw=i); in = n pairs (a,b) */
int pt_math_probe(pt_ctx*, int which, const double* in, uint32_t n, double* out);

#ifdef __cplusplus
}
#endif
#endif
