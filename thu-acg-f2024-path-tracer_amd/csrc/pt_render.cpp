// Context management and the wavefront render loop (host side of Camera::render,
// camera.rs:79-126): size the path pool, launch init -> {extend, shade}* -> resolve on one HIP
// stream, poll the live-slot counter every few iterations, report per-kernel HIP-event times.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/pt_amd.h"
#include "pt_kernels.h"
#include "pt_scene.h"

using namespace pt;
using namespace pt::host;

extern "C" const char* pt_last_error(void) { return pt::last_error(); }
extern "C" int pt_set_error_message(const char* msg) { return set_error(msg); }

extern "C" int pt_ctx_create(int device, pt_ctx** out) {
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return set_error("pt_ctx_create: no HIP device available — this library has no CPU fallback");
    if (device < 0 || device >= n) return set_error("pt_ctx_create: device index out of range");
    if (!hip_ok(hipSetDevice(device), "hipSetDevice")) return -1;
    hipDeviceProp_t prop;
    if (!hip_ok(hipGetDeviceProperties(&prop, device), "hipGetDeviceProperties")) return -1;
    pt_ctx* c = new pt_ctx();
    c->device = device;
    c->n_cus = prop.multiProcessorCount;
    c->name = std::string(prop.name) + " (" + prop.gcnArchName + ")";
    if (!hip_ok(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking), "hipStreamCreate")) {
        delete c;
        return -1;
    }
    *out = c;
    return 0;
}
extern "C" void pt_ctx_destroy(pt_ctx* c) {
    if (!c) return;
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}
extern "C" int pt_device_name(pt_ctx* c, char* buf, uint32_t n) {
    if (!c || !buf || n == 0) return set_error("pt_device_name: bad arguments");
    strncpy(buf, c->name.c_str(), n - 1);
    buf[n - 1] = 0;
    return 0;
}
extern "C" pt_scene* pt_scene_create(pt_ctx* c) {
    if (!c) {
        set_error("pt_scene_create: null context");
        return nullptr;
    }
    pt_scene* s = new pt_scene();
    s->ctx = c;
    return s;
}
extern "C" void pt_scene_destroy(pt_scene* s) { delete s; }
extern "C" pt_ctx* pt_scene_ctx(pt_scene* s) { return s ? s->ctx : nullptr; }
extern "C" int pt_find_registered_image(pt_scene* s, const char* name) {
    auto it = s->images.find(name);
    return it == s->images.end() ? -1 : it->second;
}

// Camera::init camera.rs:51-77 (host, once per render)
namespace {
struct CamDerived {
    D3 forward, right, up, center, pixel00, pixel_du, pixel_dv;
    uint32_t height;
};
int derive_camera(const pt_camera* c, CamDerived& d) {
    if (c->image_width == 0 || !(c->aspect_ratio > 0.0)) return set_error("camera: image_width and aspect_ratio must be positive");
    d.height = (uint32_t)((double)c->image_width / c->aspect_ratio);
    if (d.height == 0) return set_error("camera: image height is zero");
    d.center = d3(c->look_from);
    double theta = c->vfov * (PI / 180.0);   // f64::to_radians
    double h = std::tan(theta / 2.0);
    double viewport_height = 2.0 * h * c->focal_length;
    double viewport_width = viewport_height * ((double)c->image_width / (double)d.height);
    d.forward = normalize(d3(c->look_from) - d3(c->look_at));
    d.right = normalize(cross(d3(c->vup), d.forward));
    d.up = cross(d.forward, d.right);
    D3 viewport_u = d.right * viewport_width;
    D3 viewport_v = d.up * -viewport_height;
    d.pixel_du = viewport_u / (double)c->image_width;
    d.pixel_dv = viewport_v / (double)d.height;
    D3 upperleft = d.center - (d.forward * c->focal_length) - (viewport_u / 2.0) - (viewport_v / 2.0);
    d.pixel00 = upperleft + (d.pixel_du + d.pixel_dv) * 0.5;
    return 0;
}
}  // namespace
extern "C" int pt_camera_init(const pt_camera* c, double out[18], uint32_t* image_height) {
    CamDerived d;
    if (derive_camera(c, d) != 0) return -1;
    const D3 v[6] = {d.forward, d.right, d.up, d.pixel00, d.pixel_du, d.pixel_dv};
    for (int i = 0; i < 6; ++i) st3(out + 3 * i, v[i]);
    *image_height = d.height;
    return 0;
}

namespace {
struct EventTimer {   // per-launch HIP-event timing, drained at the polling syncs
    struct Pending {
        hipEvent_t a, b;
        int kind;
    };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> free_list;
    double ms[3] = {0, 0, 0};
    uint64_t launches[3] = {0, 0, 0};
    bool enabled = false;
    hipEvent_t get() {
        if (!free_list.empty()) {
            hipEvent_t e = free_list.back();
            free_list.pop_back();
            return e;
        }
        hipEvent_t e;
        (void)hipEventCreate(&e);
        return e;
    }
    void begin(int kind, hipStream_t st) {
        ++launches[kind];
        if (!enabled) return;
        Pending p{get(), get(), kind};
        (void)hipEventRecord(p.a, st);
        pending.push_back(p);
    }
    void end(hipStream_t st) {
        if (!enabled) return;
        (void)hipEventRecord(pending.back().b, st);
    }
    void drain() {   // call after a stream sync
        for (auto& p : pending) {
            float t = 0;
            if (hipEventElapsedTime(&t, p.a, p.b) == hipSuccess) ms[p.kind] += t;
            free_list.push_back(p.a);
            free_list.push_back(p.b);
        }
        pending.clear();
    }
    ~EventTimer() {
        drain();
        for (auto e : free_list) (void)hipEventDestroy(e);
    }
};
}  // namespace

extern "C" int pt_render(pt_scene* s, const pt_camera* cam, uint64_t seed, uint32_t spp_begin, uint32_t spp_end, double* accum,
                         const pt_render_opts* opts_in, pt_render_stats* stats) {
    if (!s || !s->built) return set_error("pt_render: world not built (call pt_world_build)");
    if (!accum) return set_error("pt_render: null accumulator");
    if (spp_end < spp_begin) return set_error("pt_render: spp_end < spp_begin");
    pt_render_opts opts;
    memset(&opts, 0, sizeof opts);
    if (opts_in) opts = *opts_in;
    pt_ctx* ctx = s->ctx;
    if (!hip_ok(hipSetDevice(ctx->device), "hipSetDevice")) return -1;
    hipStream_t st = opts.stream ? (hipStream_t)opts.stream : ctx->stream;

    CamDerived cd;
    if (derive_camera(cam, cd) != 0) return -1;
    CamD dc;
    memset(&dc, 0, sizeof dc);
    st3(dc.center, cd.center); st3(dc.pixel00, cd.pixel00); st3(dc.pixel_du, cd.pixel_du); st3(dc.pixel_dv, cd.pixel_dv);
    double lens_radius = std::tan((cam->defocus_angle / 2.0) * (PI / 180.0)) * cam->focal_length;   // camera.rs:159
    st3(dc.dof_right, cd.right * lens_radius);
    st3(dc.dof_up, cd.up * lens_radius);
    dc.blur_strength = cam->blur_strength;
    for (int i = 0; i < 3; ++i) dc.env_color[i] = cam->env_color[i];
    {   // rand 0.8.5 UniformFloat::new_inclusive(0, 2pi): scale = (high-low)/(1-eps), nudged down if needed
        const double hi = 2.0 * PI, max_rand = 1.0 - 1.0 / 4503599627370496.0;
        double scale = hi / max_rand;
        while (!(scale * max_rand <= hi)) scale = std::nextafter(scale, 0.0);
        dc.two_pi_scale = scale;
    }
    dc.width = cam->image_width;
    dc.height = cd.height;
    dc.max_depth = cam->max_depth;
    dc.env_is_map = cam->env_is_map ? 1u : 0u;
    dc.env_tex = cam->env_tex;
    dc.n_lights = s->dev.view.n_lights;
    {   // camera.rs:159-163 with radius 0: origin = center + 0 * px + 0 * py = center bit for bit (px, py are finite), unless a
        // component of center is -0.0 (then -0 + +0 = +0): only then must the products be formed
        bool zero = true;
        for (int i = 0; i < 3; ++i)
            zero = zero && dc.dof_right[i] == 0.0 && dc.dof_up[i] == 0.0 && !(dc.center[i] == 0.0 && std::signbit(dc.center[i]));
        dc.lens_zero = zero ? 1u : 0u;
        // sphere.rs:64-66: center = p1 + (p2 - p1) * time; with p1 == p2 that is p1 + (+0) * time = p1 for every time in [0, 1)
        dc.motionless = s->motionless && !exp_env("PT_DRAW_TIME") ? 1u : 0u;
    }
    if (dc.env_is_map) {
        if (cam->env_tex < 0 || (size_t)cam->env_tex >= s->tex.size() || (s->tex[cam->env_tex].d.kind != TEX_IMAGE && s->tex[cam->env_tex].d.kind != TEX_IMAGE_F32))
            return set_error("pt_render: env_tex must be an image texture of this scene");
    }
    const uint64_t n_pixels64 = (uint64_t)dc.width * dc.height;
    if (n_pixels64 == 0 || n_pixels64 > 0x7FFFFFFFull) return set_error("pt_render: bad image size");
    const uint32_t n_pixels = (uint32_t)n_pixels64;
    const uint32_t spp = spp_end - spp_begin;

    // pool sizing. slots_per_pixel = 0 (default): DYNAMIC work assignment — a fixed pool that fills
    // the machine several times over; finished paths pull the next (pixel, sample) from a global
    // counter. slots_per_pixel = k >= 1: STATIC ownership (deterministic; k = 1 is the reference's
    // exact per-pixel sample order).
    uint32_t k = opts.slots_per_pixel;
    if (k == 0)   // an explicit option wins over the experiment switch
        if (const char* e = exp_env("PT_SLOTS_PER_PIXEL")) k = (uint32_t)atoi(e);
    const bool dynamic = k == 0;
    const uint32_t tiles_x = (dc.width + 7) / 8, tiles_y = (dc.height + 7) / 8;
    const uint64_t n_tile_pixels64 = (uint64_t)tiles_x * tiles_y * 64;
    if (n_tile_pixels64 > 0x7FFFFFFFull) return set_error("pt_render: image too large");
    const uint64_t total_work = dynamic ? n_tile_pixels64 * spp : (uint64_t)n_pixels * spp;
    uint64_t n_slots64;
    if (dynamic) {
        // Resident paths: enough that per-launch fixed costs and kernel tails amortise (16.8M slots are 9% faster than
        // 4.2M on the 4000-spp frame, 33.6M another 2%), few enough that the frame's end — when the sample budget is
        // handed out and slots die — stays short: between 16K and 128K slots per CU (3.5 GB of path state at 33.6M).
        // [r3] about 64 samples per slot (was 128; 32 gained another 5-15 % on scene 6 below 500 spp but lost 4 % on scene 5 at 4K): re-measured for the sample ranges ONE RANK of an 8-GPU frame renders — scene 6
        // FHD @ 500 spp: 4.2 M slots 489 ms, 8.4 M (the old rule's choice) 426, 16.8 M 407.5, 33.6 M 409; @ 1000 spp: 8.4 M 846,
        // 16.8 M (old) 785.5, 33.6 M 779 — the long, thin end of a frame costs less than running the whole frame on a small pool.
        // [r3, last afternoon] re-measured once more with the 512-thread K3, its 8192-slot windows and the half windows at the end of K2's
        // queue (every launch's END costs less, so deeper pools pay): about 30 samples per slot and up to 512 K slots per CU — scene 6
        // FHD @ 4000 spp: 33.6 M slots 2905, 67 M 2940, 134 M 2984, 268 M 2957 Msamples/s; @ 2000: 2886 / 2903 / 2923; @ 1000: 33.6 M 2822,
        // 67 M 2836, 134 M 2752; @ 500: 16.8 M 2688, 33.6 M 2750, 67 M 2694; @ 250: 8.4 M 2484, 16.8 M 2574; scene 3 1920x1920 @ 4000:
        // 1314 / 1328 / 1359; scene 5 4K @ 1000: 4355 / 4431 / 4499 (profiles/r03_pool_sweep.txt). 134 M slots are 14 GB of path records.
        uint64_t per_cu = 16384;   // a power of two (the tile-ordered work items and the 64 counter shards divide it evenly)
        while (per_cu < 524288 && per_cu * 2 * (uint64_t)std::max(1, ctx->n_cus) * 30 <= total_work) per_cu *= 2;
        // one more doubling (268 M slots, 28 GB) only from 48 samples per slot: scene 3 1920x1920 @ 4000 spp (55 per slot) 1335 -> 1362,
        // while at 31 per slot scene 6 FHD @ 4000 loses 0.7 % and scene 5 4K @ 1000 0.5 % (initialising and compacting the pool costs 58 ms there)
        if (per_cu == 524288 && per_cu * 2 * (uint64_t)std::max(1, ctx->n_cus) * 48 <= total_work) per_cu *= 2;
        uint64_t target = (uint64_t)ctx->n_cus * per_cu;
        if (const char* e = exp_env("PT_POOL_SLOTS")) {
            target = strtoull(e, nullptr, 10);
            if (target == 0) return set_error("pt_render: PT_POOL_SLOTS must be positive");
        }
        n_slots64 = std::min<uint64_t>(target, std::max<uint64_t>(total_work, 1));
    } else {
        if (k > spp) k = spp;
        if (k == 0) k = 1;
        while ((uint64_t)k * n_pixels > 0x40000000ull && k > 1) --k;
        n_slots64 = (uint64_t)k * n_pixels;
    }
    if (n_slots64 > 0x7FFFFFC0ull) return set_error("pt_render: image too large for the path pool");
    const uint32_t n_slots = (uint32_t)n_slots64;
    // one allocation: the two record arrays (RayRec, PathRec), the static mode's f64 arrays, the two u32 state arrays
    const size_t n_al = ((size_t)n_slots + 8191) & ~(size_t)8191;   // whole windows: 2048 slots (k_extend2, k_shade) / 4096 (k_shade with 512 threads)
    const size_t n_f64 = dynamic ? 0 : 6;   // the per-slot sample sums and radiances exist in the static mode only (the dynamic mode adds into the frame)
    const size_t bytes = n_al * (sizeof(RayRec) + sizeof(PathRec) + n_f64 * sizeof(double) + 2 * sizeof(uint32_t));
    if (bytes > s->pool_bytes) {
        if (s->pool_mem) (void)hipFree(s->pool_mem);
        s->pool_mem = nullptr;
        s->pool_bytes = 0;
        if (!hip_ok(hipMalloc(&s->pool_mem, bytes), "hipMalloc(path pool)")) return -1;
        s->pool_bytes = bytes;
    }
    if (!s->d_counters) {
        if (!hip_ok(hipMalloc((void**)&s->d_counters, sizeof(CountersD)), "hipMalloc(counters)")) return -1;
        if (!hip_ok(hipHostMalloc((void**)&s->h_counters, sizeof(CountersD), hipHostMallocDefault), "hipHostMalloc(counters)")) return -1;
    }
    PoolD pool;
    memset(&pool, 0, sizeof pool);
    {
        char* m = (char*)s->pool_mem;   // hipMalloc memory is 256-B aligned; records first (64-B aligned)
        pool.ray = (RayRec*)m; m += n_al * sizeof(RayRec);
        pool.path = (PathRec*)m; m += n_al * sizeof(PathRec);
        double* d = (double*)m;
        double** f64s[6] = {&pool.ax, &pool.ay, &pool.az, &pool.rx, &pool.ry, &pool.rz};
        for (auto p : f64s) { *p = n_f64 ? d : nullptr; d += n_f64 ? n_al : 0; }
        uint32_t* u = (uint32_t*)d;
        uint32_t** u32s[2] = {&pool.hit_prim, &pool.bounce};
        for (auto p : u32s) { *p = u; u += n_al; }
    }
    pool.n_slots = n_slots;
    pool.n_alloc = (uint32_t)n_al;
    pool.n_pixels = n_pixels;
    pool.k = dynamic ? 0u : k;
    pool.spp_begin = spp_begin;
    pool.spp_end = spp_end;
    pool.dynamic = dynamic ? 1u : 0u;
    pool.defer_regen = (dynamic && !exp_env("PT_NO_DEFER_REGEN")) ? 1u : 0u;
    pool.compact = (dc.motionless && !exp_env("PT_NO_COMPACT_RECORDS")) ? 1u : 0u;
    pool.total_work = total_work;
    pool.width = dc.width;
    pool.height = dc.height;
    pool.tiles_x = tiles_x;
    pool.n_tile_pixels = (uint32_t)n_tile_pixels64;

    // accumulator on the device (freed on every return path when it is ours)
    double* d_accum = accum;
    const size_t accum_bytes = (size_t)n_pixels * 3 * sizeof(double);
    struct AccumGuard {
        double* p = nullptr;
        ~AccumGuard() { if (p) (void)hipFree(p); }
    } own_accum;
    if (!opts.accum_on_device) {
        if (!hip_ok(hipMalloc((void**)&d_accum, accum_bytes), "hipMalloc(accum)")) return -1;
        own_accum.p = d_accum;
        if (!hip_ok(hipMemsetAsync(d_accum, 0, accum_bytes, st), "hipMemset(accum)")) return -1;
    } else if (opts.overwrite) {
        if (!hip_ok(hipMemsetAsync(d_accum, 0, accum_bytes, st), "hipMemset(accum)")) return -1;
    }

    // persistent grids: resident blocks per CU x CUs
    int mult = 1;
    if (const char* e = exp_env("PT_GRID_MULT")) mult = std::max(1, atoi(e));
    int shade_variant = 42;   // k_shade<sort, min waves/SIMD>: sort*10 + waves (12 = windowed material sort with 256 threads / 2048-slot windows, 2 = plain;
                              // [r3] 22 = the same with 512 threads / 4096-slot windows: K3 -2 % on scenes 6, 3 and 5; 32 = 8192-slot windows: another
                              // -1.6 % on scene 6's 33.6 M-slot pool, +2.5 % on scene 5's 16.8 M; 42 = per launch, 32 while the pool holds >= 16 such
                              // windows per block launched, else 22)
    if (const char* e = exp_env("PT_SHADE_VARIANT")) shade_variant = atoi(e);
    uint32_t wide_window_min = 16;
    if (const char* e = exp_env("PT_WIDE_WINDOW_MIN")) wide_window_min = (uint32_t)std::max(1, atoi(e));
    // K2 variant: two-phase kernel when there are meshes to defer and its LDS stack covers the scene's BVHs, else the
    // batch kernel. Experiment switches: PT_K2=batch forces the batch kernel; PT_EXT2 = stack*10 + blocks per CU picks
    // the instantiation. extend_code: -1 = batch, -(stack*10 + blocks) = two-phase.
    auto extend2_code = [&]() -> int {
        const int need = (int)s->stack_need_extend2;
        if (need > 32) return 0;
        // four blocks per CU where the LDS allows it (stacks of 16 and 20 entries): the kernel then runs at 128 registers with
        // 64 B of spills per lane and is still 7.5 % faster than at three blocks and 149 registers (round 2; in round 1, at
        // 166 registers, the same bound meant 168 B of spills and lost 11 %)
        // [r3, last] stacks of <= 16 entries: blocks of 128 threads over 1024-slot windows (code 2164, eight blocks per CU) — with the
        // queue's end in half windows the smaller block's shorter barrier waits win on every pool size: K2 -1.0 % (16.8 M slots), -1.8 %
        // (67 M, 134 M) against 256 threads; 64 threads: +1 % / -2.0 % / -2.5 % (worse on shallow pools); 512 threads: +5 %
        int code = need <= 16 ? 2164 : need <= 20 ? 204 : need <= 24 ? 243 : need <= 28 ? 283 : 323;
        if (const char* e = exp_env("PT_EXT2")) {
            const int c = atoi(e);
            if (c / 10 >= need && (c == 163 || c == 164 || c == 204 || c == 243 || c == 283 || c == 323)) code = c;
            if ((c == 1164 || c == 2164 || c == 8164) && need <= 16) code = c;
        }
        return code;
    };
    int extend_code = (s->n_mesh_entries > 0 && extend2_code() != 0) ? -extend2_code() : -1;
    if (const char* e = exp_env("PT_K2")) {
        if (!strcmp(e, "batch")) extend_code = -1;
        else if (!strcmp(e, "twophase") && extend2_code() != 0) extend_code = -extend2_code();
    }
    const int blocks_extend = kernel_occupancy_blocks(0, extend_code == -1 && s->dev.view.tlas_flat ? (s->dev.view.flat_pairs ? -3 : -2) : extend_code), blocks_shade = kernel_occupancy_blocks(1, shade_variant, s->dev.view.n_lights != 0u);
    const int grid_extend = ctx->n_cus * blocks_extend * mult, grid_shade = ctx->n_cus * blocks_shade * mult;

    pool.accum = d_accum;
    // dynamic mode: the kernels add into channel planes in work-item (tile) order (PoolD::accum_tiled); k_detile adds them to d_accum
    const bool tiled = dynamic && !exp_env("PT_ACCUM_LINEAR");
    if (tiled) {
        const size_t tb = (size_t)n_tile_pixels64 * 3 * sizeof(double);
        if (tb > s->tile_accum_bytes) {
            if (s->tile_accum) (void)hipFree(s->tile_accum);
            s->tile_accum = nullptr;
            s->tile_accum_bytes = 0;
            if (!hip_ok(hipMalloc((void**)&s->tile_accum, tb), "hipMalloc(tiled accumulator)")) return -1;
            s->tile_accum_bytes = tb;
        }
        if (!hip_ok(hipMemsetAsync(s->tile_accum, 0, tb, st), "hipMemset(tiled accumulator)")) return -1;
        pool.accum = s->tile_accum;
        pool.accum_tiled = 1u;
    }
    pool.inv_width = 1.0 / (double)dc.width;
    CountersD init_cnt;
    memset(&init_cnt, 0, sizeof init_cnt);
    init_cnt.alive = spp == 0 ? 0 : n_slots;   // every slot starts with one sample (k <= spp / n_slots <= total_work)
    for (uint32_t sh = 0; sh < WORK_SHARDS; ++sh) {   // dynamic mode: items 0 .. n_slots-1 were handed out by k_init
        const uint64_t row = (uint64_t)WORK_SHARDS * 64, rows = n_slots / row, rem = n_slots % row;
        const uint64_t part = rem > (uint64_t)sh * 64 ? std::min<uint64_t>(rem - (uint64_t)sh * 64, 64) : 0;
        init_cnt.work[sh].next = rows * 64 + part;
    }
    if (!hip_ok(hipMemcpyAsync(s->d_counters, &init_cnt, sizeof init_cnt, hipMemcpyHostToDevice, st), "hipMemcpy(counters)")) return -1;

    EventTimer timer;
    timer.enabled = opts.profile != 0;
    auto t0 = std::chrono::steady_clock::now();
    (void)hipStreamSynchronize(st);
    t0 = std::chrono::steady_clock::now();

    timer.begin(2, st);
    launch_init(dc, pool, seed, grid_shade, st);
    timer.end(st);
    uint64_t iterations = 0;
    const uint64_t per_slot = dynamic ? (total_work + n_slots - 1) / std::max<uint64_t>(n_slots, 1) + 1 : (spp + k - 1) / k;
    const uint64_t max_iterations = per_slot * ((uint64_t)std::max(1u, dc.max_depth) + 1) + 4;   // + 1: a parked slot idles one iteration
    uint32_t poll_every = 8;
    const bool compact_ok = dynamic && !exp_env("PT_NO_COMPACT_POOL");
    uint32_t compactions = 0;
    uint64_t compact_num = 1, compact_den = 2;      // compact when live <= num/den of the slots still covered (25 % .. 85 % measured level: within 0.5 %)
    uint32_t poll_cap = 8;
    if (const char* e = exp_env("PT_COMPACT_AT")) { compact_num = (uint64_t)std::max(1, atoi(e)); compact_den = 100; }   // per cent
    if (const char* e = exp_env("PT_POLL_CAP")) poll_cap = (uint32_t)std::max(1, atoi(e));
    bool alive = spp != 0 && dc.max_depth != 0;
    if (spp != 0 && dc.max_depth == 0) {
        // max_depth = 0: trace() returns zero radiance for every sample (camera.rs:177); nothing to launch
        alive = false;
    }
    while (alive) {
        for (uint32_t i = 0; i < poll_every; ++i) {
            timer.begin(0, st);
            launch_extend(s->dev.view, pool, s->d_counters, grid_extend, extend_code, st);
            timer.end(st);
            timer.begin(1, st);
            launch_shade(s->dev.view, dc, pool, s->d_counters, seed, grid_shade, shade_variant, st, wide_window_min);
            timer.end(st);
            ++iterations;
        }
        if (!hip_ok(hipMemcpyAsync(s->h_counters, s->d_counters, sizeof(CountersD), hipMemcpyDeviceToHost, st), "hipMemcpy(counters)")) return -1;
        if (!hip_ok(hipStreamSynchronize(st), "hipStreamSynchronize(render)")) return -1;
        timer.drain();
        const uint64_t n_alive = s->h_counters->alive;
        alive = n_alive != 0;
        if (alive && iterations > max_iterations + 128) return set_error("pt_render: iteration bound exceeded (internal error)");
        if (poll_every < poll_cap) poll_every *= 2;     // (a poll is a pipeline drain of some tens of microseconds: every 8 iterations costs <= 0.5 %)
        // the frame's end: once half of the slots still covered are dead the survivors move to the front and the launches shrink with them
        // (k_compact_scan / k_compact_move). The count is the last poll's — stale only towards MORE live slots, which errs on the safe side.
        if (compact_ok && alive && n_alive * compact_den <= (uint64_t)pool.n_alloc * compact_num && pool.n_alloc > 4 * 8192u) {
            const uint32_t new_end = (uint32_t)((n_alive + 8191) & ~(uint64_t)8191);
            const uint32_t cap = new_end;                                  // holes and movers are both at most the live count
            const size_t words = 2 * (size_t)cap + 2;
            if (words > s->compact_scratch_words) {
                if (s->compact_scratch) (void)hipFree(s->compact_scratch);
                s->compact_scratch = nullptr;
                s->compact_scratch_words = 0;
                if (!hip_ok(hipMalloc((void**)&s->compact_scratch, words * sizeof(uint32_t)), "hipMalloc(compaction lists)")) return -1;
                s->compact_scratch_words = words;
            }
            timer.begin(2, st);
            launch_compact(pool, new_end, s->compact_scratch, s->compact_scratch + cap, s->compact_scratch + 2 * (size_t)cap, cap, ctx->n_cus * 8, st);
            timer.end(st);
            pool.n_alloc = new_end;
            pool.n_slots = std::min(pool.n_slots, new_end);
            ++compactions;
        }
    }
    if (!dynamic || tiled) {
        timer.begin(2, st);
        if (tiled) launch_detile(pool, d_accum, ctx->n_cus * 8, st);
        else launch_resolve(pool, d_accum, ctx->n_cus * 8, st);
        timer.end(st);
    }
    if (!hip_ok(hipMemcpyAsync(s->h_counters, s->d_counters, sizeof(CountersD), hipMemcpyDeviceToHost, st), "hipMemcpy(counters)")) return -1;
    if (!hip_ok(hipStreamSynchronize(st), "hipStreamSynchronize(resolve)")) return -1;
    timer.drain();
    auto t1 = std::chrono::steady_clock::now();
    if (!hip_ok(hipGetLastError(), "kernel launch")) return -1;

    if (!opts.accum_on_device) {
        if (opts.overwrite) {
            if (!hip_ok(hipMemcpy(accum, d_accum, accum_bytes, hipMemcpyDeviceToHost), "hipMemcpy(accum)")) return -1;
        } else {
            std::vector<double> tmp((size_t)n_pixels * 3);
            if (!hip_ok(hipMemcpy(tmp.data(), d_accum, accum_bytes, hipMemcpyDeviceToHost), "hipMemcpy(accum)")) return -1;
            for (size_t i = 0; i < tmp.size(); ++i) accum[i] += tmp[i];
        }
    }
    if (exp_env("PT_PROF")) {   // diagnostic builds (-DPT_STAMPS): wave-cycle sums per k_shade class
        static const char* names[N_CLASSES + 1] = {"miss", "diffuse", "metal", "glass", "principled", "light", "sheen", "clearcoat", "mix", "idle", "dead", "WINDOW"};
        for (uint32_t c = 0; c <= N_CLASSES; ++c) {
            const unsigned long long* p = s->h_counters->prof[c];
            if (p[0] && c == CLASS_DEAD)
                fprintf(stderr, "[pt prof] K2 window  n %10llu  phase A %8.0f  barrier %8.0f  phase B %8.0f  barrier %8.0f  candidates %6.1f (cycles per window and wave)\n", p[0],
                        (double)p[1] / p[0], (double)p[2] / p[0], (double)p[3] / p[0], (double)p[4] / p[0], (double)p[5] / p[0]);
            else if (p[0] && c < N_CLASSES)
                fprintf(stderr, "[pt prof] %-10s n %10llu  body %8.0f = hit %7.0f + env/tex %7.0f + direction %7.0f + pdf/eval/ray %7.0f  dequeue %8.0f  regen+store %8.0f  whole %8.0f (cycles per wave-group)\n",
                        names[c], p[0], (double)p[2] / p[0], (double)p[1] / p[0], (double)p[6] / p[0], (double)p[7] / p[0], (double)(p[2] - p[1] - p[6] - p[7]) / p[0],
                        (double)p[3] / p[0], (double)p[4] / p[0], (double)p[5] / p[0]),
                fprintf(stderr, "[pt prof] %-10s   lanes per group: live %5.1f  on a surface %5.1f  with a next direction %5.1f  regenerated %5.1f\n", names[c], (double)p[8] / p[0],
                        (double)p[9] / p[0], (double)p[10] / p[0], (double)p[11] / p[0]);
            else if (p[0])
                fprintf(stderr, "[pt prof] WINDOW     n %10llu  sort %8.0f  shade %8.0f  barrier wait %8.0f  groups %8.0f  record wait %8.0f (cycles per window and wave)\n", p[0],
                        (double)p[1] / p[0], (double)p[2] / p[0], (double)p[3] / p[0], (double)p[4] / p[0], (double)p[5] / p[0]);
        }
    }
    if (stats) {
        memset(stats, 0, sizeof *stats);
        stats->samples = s->h_counters->samples;
        stats->segments = s->h_counters->segments;
        stats->iterations = iterations;
        stats->n_slots = n_slots;
        stats->slots_per_pixel = dynamic ? 0u : k;
        stats->ms_total = std::chrono::duration<double, std::milli>(t1 - t0).count();
        stats->ms_extend = timer.ms[0];
        stats->ms_shade = timer.ms[1];
        stats->ms_other = timer.ms[2];
        stats->launches_extend = timer.launches[0];
        stats->launches_shade = timer.launches[1];
        stats->extend_variant = extend_code <= -100 ? 0u : 1u;
        stats->shade_variant = (uint32_t)shade_variant;
        stats->blocks_extend = (uint32_t)grid_extend;
        stats->blocks_shade = (uint32_t)grid_shade;
        stats->compactions = compactions;
        stats->n_alloc_end = pool.n_alloc;
    }
    return 0;
}

extern "C" int pt_resolve_u8(pt_ctx* ctx, const double* accum, uint32_t n_pixels, uint32_t total_spp, uint8_t* rgb8) {
    if (!ctx) return set_error("pt_resolve_u8: null context");
    if (!hip_ok(hipSetDevice(ctx->device), "hipSetDevice")) return -1;
    const uint32_t n = n_pixels * 3;
    double* d_in = nullptr;
    uint8_t* d_out = nullptr;
    bool ok = hip_ok(hipMalloc((void**)&d_in, (size_t)n * sizeof(double)), "hipMalloc") && hip_ok(hipMalloc((void**)&d_out, n), "hipMalloc") &&
              hip_ok(hipMemcpyAsync(d_in, accum, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream), "hipMemcpy");
    if (ok) {
        launch_quantise(d_in, n, 1.0 / (double)total_spp, d_out, ctx->stream);   // pixel_sample_scale camera.rs:53
        ok = hip_ok(hipMemcpyAsync(rgb8, d_out, n, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpy") &&
             hip_ok(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
    }
    if (d_in) (void)hipFree(d_in);
    if (d_out) (void)hipFree(d_out);
    return ok ? 0 : -1;
}

extern "C" int pt_intersect(pt_scene* s, const double* rays, uint32_t n, double* out) {
    if (!s || !s->built) return set_error("pt_intersect: world not built");
    pt_ctx* ctx = s->ctx;
    if (!hip_ok(hipSetDevice(ctx->device), "hipSetDevice")) return -1;
    double *d_r = nullptr, *d_o = nullptr;
    bool ok = hip_ok(hipMalloc((void**)&d_r, (size_t)n * 7 * sizeof(double) + 8), "hipMalloc") &&
              hip_ok(hipMalloc((void**)&d_o, (size_t)n * 15 * sizeof(double) + 8), "hipMalloc") &&
              hip_ok(hipMemcpyAsync(d_r, rays, (size_t)n * 7 * sizeof(double), hipMemcpyHostToDevice, ctx->stream), "hipMemcpy");
    if (ok) {
        launch_probe(s->dev.view, d_r, n, d_o, ctx->stream);
        ok = hip_ok(hipMemcpyAsync(out, d_o, (size_t)n * 15 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream), "hipMemcpy") &&
             hip_ok(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
    }
    if (d_r) (void)hipFree(d_r);
    if (d_o) (void)hipFree(d_o);
    return ok ? 0 : -1;
}

extern "C" int pt_math_probe(pt_ctx* ctx, int which, const double* in, uint32_t n, double* out) {
    if (!ctx) return set_error("pt_math_probe: null context");
    if (!hip_ok(hipSetDevice(ctx->device), "hipSetDevice")) return -1;
    double *d_i = nullptr, *d_o = nullptr;
    bool ok = hip_ok(hipMalloc((void**)&d_i, (size_t)n * 2 * sizeof(double) + 8), "hipMalloc") &&
              hip_ok(hipMalloc((void**)&d_o, (size_t)n * sizeof(double) + 8), "hipMalloc") &&
              hip_ok(hipMemcpyAsync(d_i, in, (size_t)n * 2 * sizeof(double), hipMemcpyHostToDevice, ctx->stream), "hipMemcpy");
    if (ok) {
        launch_math_probe(which, d_i, n, d_o, ctx->stream);
        ok = hip_ok(hipMemcpyAsync(out, d_o, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream), "hipMemcpy") &&
             hip_ok(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
    }
    if (d_i) (void)hipFree(d_i);
    if (d_o) (void)hipFree(d_o);
    return ok ? 0 : -1;
}
