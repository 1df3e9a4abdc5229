// Host-callable launchers of the HIP kernels in pt_kernels.hip.
#pragma once
#include <hip/hip_runtime_api.h>

#include "pt_types.h"

namespace pt {
void launch_init(const CamD& cam, const PoolD& pool, uint64_t seed, int max_blocks, hipStream_t st);
void launch_extend(const SceneD& sc, const PoolD& pool, CountersD* cnt, int max_blocks, int code, hipStream_t st);
// wide_window_min (variant 42): 8192-slot windows while the pool holds at least that many of them per block launched, 4096-slot ones below
void launch_shade(const SceneD& sc, const CamD& cam, const PoolD& pool, CountersD* cnt, uint64_t seed, int max_blocks, int variant,
                  hipStream_t st, uint32_t wide_window_min = 16);
void launch_resolve(const PoolD& pool, double* accum, int max_blocks, hipStream_t st);
// the frame's end (dynamic mode): live slots beyond new_end move into dead slots below it; holes / movers: scratch lists of `cap` entries, counts: 2 words
void launch_compact(const PoolD& pool, uint32_t new_end, uint32_t* holes, uint32_t* movers, uint32_t* counts, uint32_t cap, int max_blocks, hipStream_t st);
void launch_detile(const PoolD& pool, double* accum, int max_blocks, hipStream_t st);
void launch_quantise(const double* accum, uint32_t n, double scale, uint8_t* rgb8, hipStream_t st);
void launch_probe(const SceneD& sc, const double* rays, uint32_t n, double* out, hipStream_t st);
void launch_math_probe(int which, const double* in, uint32_t n, double* out, hipStream_t st);
// K2 variant code (`code` of launch_extend / `variant` of kernel_occupancy_blocks): -1 = batch kernel (-2 asks
// kernel_occupancy_blocks for its flat-top-level instantiation), -(stack*10 + blocks) = two-phase kernel
// k_extend2<stack, blocks> for stack in {16, 20, 24}.
int kernel_occupancy_blocks(int which, int variant, bool lights = false);   // lights: k_shade's instantiation for scenes with a lights list   // 0 = extend, 1 = shade; resident blocks per CU
}  // namespace pt
