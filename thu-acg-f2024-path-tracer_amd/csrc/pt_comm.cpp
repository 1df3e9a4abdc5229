// Multi-GPU behind the C ABI: one process per GPU, spp sharding, ONE RCCL reduce of the f64 W*H*3 SUM accumulators
// over xGMI (camera.rs:106-108 only sums the samples of a pixel, so sample ranges add up). RCCL is called
// directly from /opt/rocm — no torch, no host bounce: every rank renders into a device accumulator on its context's
// stream and ncclReduce runs on that same stream right behind the last kernel.
//
// Rendezvous: the ranks of one launch agree on a file path (bench.py derives it from MASTER_PORT and the launcher's
// pid). Rank 0 calls ncclGetUniqueId and publishes the 128 bytes there (temp name + rename: readers never see a
// partial file); the others poll for it. That is the only out-of-band exchange; everything after is RCCL.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pt_amd.h"
#include "pt_scene.h"

using namespace pt;

struct pt_comm {
    pt_ctx* ctx = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    double* d_scratch = nullptr;     // small device buffer for host-value collectives
    double* d_accum = nullptr;       // frame accumulator of pt_render_multi (grown on demand)
    size_t accum_bytes = 0;
    double* h_stage = nullptr;       // rank 0: pinned landing buffer of the reduced frame (grown on demand)
    size_t stage_bytes = 0;
    bool aborted = false;            // ncclCommAbort was called after a local failure: every later collective fails at once
};

// RCCL prints a version banner ("RCCL version : ...", five lines) on STDOUT when the first communicator of a process is
// created. A host that reports its result on stdout (bench.py prints ONE JSON line) must not find it there: while the
// communicator is being created stdout is pointed at stderr, and whatever the C library buffered is flushed there.
struct StdoutToStderr {
    int saved;
    StdoutToStderr() {
        fflush(stdout);
        saved = dup(1);
        if (saved >= 0) (void)dup2(2, 1);
    }
    ~StdoutToStderr() {
        fflush(stdout);
        if (saved >= 0) {
            (void)dup2(saved, 1);
            close(saved);
        }
    }
};

static bool nccl_ok(ncclResult_t r, const char* what) {
    if (r == ncclSuccess) return true;
    set_error(std::string(what) + ": " + ncclGetErrorString(r));
    return false;
}

extern "C" void pt_shard_range(uint32_t spp, int rank, int world, uint32_t* lo, uint32_t* hi) {
    // contiguous, disjoint, near-equal slices of [0, spp); sample indices are global (they key the RNG), so the union over
    // the ranks is exactly the single-GPU sample set
    if (world < 1) world = 1;
    if (rank < 0) rank = 0;
    if (rank >= world) rank = world - 1;
    const uint32_t base = spp / (uint32_t)world, rem = spp % (uint32_t)world;
    const uint32_t r = (uint32_t)rank;
    const uint32_t b = r * base + (r < rem ? r : rem);
    *lo = b;
    *hi = b + base + (r < rem ? 1u : 0u);
}

// Publishes (rank 0) or fetches (other ranks) `n` bytes through the file `path`. Host-only: used for the RCCL unique id,
// testable without a GPU.
extern "C" int pt_bootstrap_exchange(const char* path, int rank, void* bytes, uint32_t n, double timeout_s) {
    if (!path || !bytes || n == 0) return set_error("pt_bootstrap_exchange: bad arguments");
    if (rank == 0) {
        const std::string tmp = std::string(path) + ".tmp";
        FILE* f = fopen(tmp.c_str(), "wb");
        if (!f) return set_error("pt_bootstrap_exchange: cannot create " + tmp);
        const bool ok = fwrite(bytes, 1, n, f) == n;
        fclose(f);
        if (!ok || rename(tmp.c_str(), path) != 0) return set_error(std::string("pt_bootstrap_exchange: cannot publish ") + path);
        return 0;
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        if (FILE* f = fopen(path, "rb")) {
            const size_t got = fread(bytes, 1, n, f);
            fclose(f);
            if (got == n) return 0;
        }
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s)
            return set_error(std::string("pt_bootstrap_exchange: timed out waiting for ") + path);
        std::this_thread::sleep_for(std::chrono::milliseconds(5));
    }
}

extern "C" int pt_comm_create(pt_ctx* ctx, int rank, int world, const char* id_path, double timeout_s, pt_comm** out) {
    if (out) *out = nullptr;
    if (!ctx || !out) return set_error("pt_comm_create: bad arguments");
    if (world < 1 || rank < 0 || rank >= world) return set_error("pt_comm_create: rank/world out of range");
    if (world > 1 && (!id_path || !*id_path)) return set_error("pt_comm_create: a rendezvous path is needed for world > 1");
    if (!hip_ok(hipSetDevice(ctx->device), "hipSetDevice")) return -1;
    StdoutToStderr quiet;   // until this function returns
    // what travels through the file: a header naming this library and the launch's world size, then the id. A reader that finds
    // anything else there (a file some other program or an earlier launch of another size left under the same name) keeps polling
    // until rank 0 has replaced it (rename is atomic) or the timeout strikes.
    struct IdFile {
        char magic[8];
        uint32_t world, reserved;
        ncclUniqueId id;
    } msg;
    static const char MAGIC[8] = {'P', 'T', 'A', 'M', 'D', 'I', 'D', '1'};
    memset(&msg, 0, sizeof msg);
    memcpy(msg.magic, MAGIC, 8);
    msg.world = (uint32_t)world;
    if (rank == 0 && !nccl_ok(ncclGetUniqueId(&msg.id), "ncclGetUniqueId")) return -1;
    if (world > 1) {
        const double limit = timeout_s > 0 ? timeout_s : 120.0;
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            const double left = limit - std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (pt_bootstrap_exchange(id_path, rank, &msg, (uint32_t)sizeof msg, left > 0.05 ? left : 0.05) != 0) return -1;
            if (rank == 0 || (memcmp(msg.magic, MAGIC, 8) == 0 && msg.world == (uint32_t)world)) break;
            if (left <= 0.05) return set_error(std::string("pt_comm_create: ") + id_path + " holds no rendezvous record of this launch (stale file?)");
            std::this_thread::sleep_for(std::chrono::milliseconds(20));
        }
    }
    const ncclUniqueId id = msg.id;
    pt_comm* c = new pt_comm();
    c->ctx = ctx;
    c->rank = rank;
    c->world = world;
    if (!nccl_ok(ncclCommInitRank(&c->comm, world, id, rank), "ncclCommInitRank") ||
        !hip_ok(hipMalloc((void**)&c->d_scratch, 64 * sizeof(double)), "hipMalloc(comm scratch)")) {
        if (c->comm) (void)ncclCommDestroy(c->comm);
        delete c;
        return -1;
    }
    *out = c;
    return 0;
}
extern "C" void pt_comm_destroy(pt_comm* c) {
    if (!c) return;
    (void)hipSetDevice(c->ctx->device);
    (void)hipStreamSynchronize(c->ctx->stream);
    if (c->d_accum) (void)hipFree(c->d_accum);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    if (c->d_scratch) (void)hipFree(c->d_scratch);
    if (c->comm) (void)ncclCommDestroy(c->comm);
    delete c;
}
extern "C" int pt_comm_rank(pt_comm* c) { return c ? c->rank : -1; }
extern "C" int pt_comm_world(pt_comm* c) { return c ? c->world : -1; }

// A rank that fails locally between two collectives must not leave its peers blocked in the next one for ever (RCCL has no
// timeout): the communicator is aborted — the peers' pending and future collectives return an error — and this handle refuses
// further use.
static int abort_comm(pt_comm* c, const std::string& why) {
    if (c->comm && !c->aborted) (void)ncclCommAbort(c->comm);
    c->comm = nullptr;
    c->aborted = true;
    return set_error(why + " (communicator aborted: the other ranks' collectives fail instead of waiting)");
}
static bool comm_usable(pt_comm* c, const char* who) {
    if (c && !c->aborted && c->comm) return true;
    set_error(std::string(who) + (c && c->aborted ? ": the communicator was aborted after an earlier failure" : ": null communicator"));
    return false;
}

// all-reduce of up to 64 host doubles (op: 0 sum, 1 max) — the bench's barrier, max-over-ranks clock and counters
extern "C" int pt_comm_allreduce_f64(pt_comm* c, double* values, uint32_t n, int op) {
    if (!comm_usable(c, "pt_comm_allreduce_f64")) return -1;
    if (!values || n == 0 || n > 64) return set_error("pt_comm_allreduce_f64: bad arguments (1..64 values)");
    hipStream_t st = c->ctx->stream;
    const bool ok = hip_ok(hipSetDevice(c->ctx->device), "hipSetDevice") &&
                    hip_ok(hipMemcpyAsync(c->d_scratch, values, n * sizeof(double), hipMemcpyHostToDevice, st), "hipMemcpy(allreduce in)") &&
                    nccl_ok(ncclAllReduce(c->d_scratch, c->d_scratch, n, ncclDouble, op == 1 ? ncclMax : ncclSum, c->comm, st), "ncclAllReduce") &&
                    hip_ok(hipMemcpyAsync(values, c->d_scratch, n * sizeof(double), hipMemcpyDeviceToHost, st), "hipMemcpy(allreduce out)") &&
                    hip_ok(hipStreamSynchronize(st), "hipStreamSynchronize(allreduce)");
    if (!ok && c->world > 1) return abort_comm(c, std::string("pt_comm_allreduce_f64: ") + pt::last_error());
    return ok ? 0 : -1;
}
extern "C" int pt_comm_barrier(pt_comm* c) {
    double one = 1.0;
    if (pt_comm_allreduce_f64(c, &one, 1, 0) != 0) return -1;
    return hip_ok(hipDeviceSynchronize(), "hipDeviceSynchronize") ? 0 : -1;
}

extern "C" int pt_render_multi(pt_scene* s, const pt_camera* cam, uint64_t seed, uint32_t spp_total, pt_comm* c, double* accum_root,
                               const pt_render_opts* opts_in, pt_render_stats* stats) {
    if (!comm_usable(c, "pt_render_multi")) return -1;
    // Everything that can fail on ONE rank only runs inside `local`; the ranks then agree on the outcome (a one-word all-reduce)
    // BEFORE the frame's reduce is posted, so that a rank whose render failed does not leave the others waiting in ncclReduce.
    size_t n = 0, bytes = 0;
    hipStream_t st = c->ctx->stream;
    pt_render_opts opts;
    memset(&opts, 0, sizeof opts);
    if (opts_in) opts = *opts_in;
    const bool overwrite = opts.overwrite != 0;
    auto local = [&]() -> int {
        if (!s || !cam) return set_error("pt_render_multi: bad arguments");
        if (pt_scene_ctx(s) != c->ctx) return set_error("pt_render_multi: the scene and the communicator belong to different contexts");
        if (c->rank == 0 && !accum_root) return set_error("pt_render_multi: rank 0 needs the output accumulator");
        if (!hip_ok(hipSetDevice(c->ctx->device), "hipSetDevice")) return -1;
        double v[18];
        uint32_t height = 0;
        if (pt_camera_init(cam, v, &height) != 0) return -1;
        n = (size_t)cam->image_width * height * 3;
        bytes = n * sizeof(double);
        if (bytes > c->accum_bytes) {
            if (c->d_accum) (void)hipFree(c->d_accum);
            c->d_accum = nullptr;
            c->accum_bytes = 0;
            if (!hip_ok(hipMalloc((void**)&c->d_accum, bytes), "hipMalloc(frame accumulator)")) return -1;
            c->accum_bytes = bytes;
        }
        if (c->rank == 0 && bytes > c->stage_bytes) {
            if (c->h_stage) (void)hipHostFree(c->h_stage);
            c->h_stage = nullptr;
            c->stage_bytes = 0;
            if (!hip_ok(hipHostMalloc((void**)&c->h_stage, bytes, hipHostMallocDefault), "hipHostMalloc(frame staging)")) return -1;
            c->stage_bytes = bytes;
        }
        if (!hip_ok(hipMemsetAsync(c->d_accum, 0, bytes, st), "hipMemset(frame accumulator)")) return -1;
        pt_render_opts o = opts;
        o.accum_on_device = 1;
        o.overwrite = 0;                 // the device accumulator was just cleared
        o.stream = (void*)st;
        uint32_t lo, hi;
        pt_shard_range(spp_total, c->rank, c->world, &lo, &hi);
        return pt_render(s, cam, seed, lo, hi, c->d_accum, &o, stats);
    };
    const int rc = local();
    if (c->world > 1) {
        const std::string own = rc != 0 ? std::string(pt::last_error()) : std::string();
        double failed = rc != 0 ? 1.0 : 0.0;
        if (pt_comm_allreduce_f64(c, &failed, 1, 1) != 0) return -1;          // (aborts the communicator when it fails itself)
        if (rc != 0) return set_error(own);
        if (failed != 0.0) return set_error("pt_render_multi: the render failed on another rank; no frame was reduced");
        // the frame's single collective: sum of the per-rank sample SUMS, in place on the root, on the render stream
        if (!nccl_ok(ncclReduce(c->d_accum, c->d_accum, n, ncclDouble, ncclSum, 0, c->comm, st), "ncclReduce"))
            return abort_comm(c, std::string("pt_render_multi: ") + pt::last_error());
    } else if (rc != 0) {
        return -1;
    }
    if (c->rank == 0) {
        // one DMA into the pinned landing buffer, then one pass over the caller's (pageable) frame
        if (!hip_ok(hipMemcpyAsync(c->h_stage, c->d_accum, bytes, hipMemcpyDeviceToHost, st), "hipMemcpy(frame)")) return -1;
        if (!hip_ok(hipStreamSynchronize(st), "hipStreamSynchronize(frame)")) return -1;
        if (overwrite) memcpy(accum_root, c->h_stage, bytes);
        else for (size_t i = 0; i < n; ++i) accum_root[i] += c->h_stage[i];
    } else if (!hip_ok(hipStreamSynchronize(st), "hipStreamSynchronize(reduce)")) {
        return -1;
    }
    return 0;
}
