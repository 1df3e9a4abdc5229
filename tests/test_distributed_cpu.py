"""N>1 path on CPU: spp sharding + one reduce of the per-rank SUM accumulators to rank 0
(gloo, world_size 2). The renderer behind the shard is the CPU oracle here (no GPU in this
container); the sharding / reduce / resolve code is the same module bench.py uses on GPUs."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, spp, out_path):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as orc
    import parallel_spp

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    s = orc.Scene()
    cam = s.build_scene(3, 24, spp)
    lo, hi = parallel_spp.shard_range(spp, rank, world)
    acc, cnt = s.render(cam, 1, lo, hi)
    t = torch.from_numpy(acc)
    parallel_spp.reduce_accum_to_root(t)                 # ONE collective: sum of the sample SUMS
    seg = torch.tensor([cnt["segments"], hi - lo], dtype=torch.int64)
    dist.all_reduce(seg)
    if rank == 0:
        np.savez(out_path, accum=t.numpy(), segments=seg[0].item(), spp=seg[1].item())
    dist.destroy_process_group()


def test_shard_range_covers_everything():
    import parallel_spp

    for spp in (0, 1, 7, 4000):
        for world in (1, 2, 3, 8):
            parts = [parallel_spp.shard_range(spp, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == spp
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1
    assert [parallel_spp.shard_range(4000, r, 8) for r in range(8)][3] == (1500, 2000)


def test_two_rank_render_equals_single_process(orc, tmp_path):
    spp = 6
    out = str(tmp_path / "r.npz")
    mp.spawn(_worker, args=(2, _free_port(), spp, out), nprocs=2, join=True)
    got = np.load(out)
    s = orc.Scene()
    cam = s.build_scene(3, 24, spp)
    ref, cnt = s.render(cam, 1, 0, spp)
    assert int(got["spp"]) == spp and int(got["segments"]) == cnt["segments"]
    # disjoint sample ranges of the same (seed, pixel, sample)-keyed streams: identical sample
    # set, only the summation order differs
    np.testing.assert_allclose(got["accum"], ref, rtol=1e-13, atol=1e-13)
    s.close()
