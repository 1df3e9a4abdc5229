"""Multi-GPU plumbing for the render path: samples shard embarrassingly (camera.rs:106-108 just
sums them), so rank r renders the contiguous sample range shard_range(spp, r, world) of every
pixel into its own SUM accumulator and ONE reduce (RCCL over xGMI on GPUs, gloo in the CPU
tests) adds the accumulators on rank 0. No other collective touches the data path."""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(spp: int, rank: int, world: int):
    """Contiguous, disjoint, near-equal slices of [0, spp). Sample indices are global (they key
    the RNG), so the union over ranks is exactly the single-GPU sample set."""
    base, rem = divmod(spp, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def reduce_accum_to_root(accum: torch.Tensor, root: int = 0):
    """Sum the per-rank W*H*3 f64 accumulators onto `root` (24.9 MB at FHD in f64 x2)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(accum, dst=root, op=dist.ReduceOp.SUM)
    return accum
