"""N>1 path on CPU (world_size 2, gloo standing in for RCCL — there is no GPU in this container).

What runs here is the host logic of the product's multi-GPU entry point, through the C ABI: pt_shard_range (which
samples a rank renders) and pt_bootstrap_exchange (the file rendezvous pt_comm_create uses to hand the RCCL unique id
from rank 0 to the others). The renderer behind the shard is the CPU oracle and the sum of the per-rank SUM
accumulators is a gloo reduce — on GPUs those two are pt_render and ncclReduce inside pt_render_multi (csrc/pt_comm.cpp;
its 1-rank form is tested on the GPU in test_gpu_parity.py)."""
import importlib
import json
import os
import socket
import subprocess
import sys

import numpy as np
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


def _worker(rank, world, port, spp, out_path, id_path):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as orc
    pt = importlib.import_module("thu-acg-f2024-path-tracer_amd")

    # the rendezvous of pt_comm_create: rank 0 publishes 128 bytes (a stand-in for the ncclUniqueId), the others wait for them
    token = bytes(range(128)) if rank == 0 else bytes(128)
    got = pt.bootstrap_exchange(id_path, rank, token, 30.0)
    assert got == bytes(range(128))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    s = orc.Scene()
    cam = s.build_scene(3, 24, spp)
    lo, hi = pt.shard_range(spp, rank, world)
    acc, cnt = s.render(cam, 1, lo, hi)
    t = torch.from_numpy(acc)
    dist.reduce(t, dst=0, op=dist.ReduceOp.SUM)          # ONE collective: sum of the sample SUMS (ncclReduce on GPUs)
    seg = torch.tensor([cnt["segments"], hi - lo], dtype=torch.int64)
    dist.all_reduce(seg)
    if rank == 0:
        np.savez(out_path, accum=t.numpy(), segments=seg[0].item(), spp=seg[1].item())
    dist.destroy_process_group()


def test_shard_range_covers_everything(pt):
    for spp in (0, 1, 7, 4000):
        for world in (1, 2, 3, 8):
            parts = [pt.shard_range(spp, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == spp
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1
    assert [pt.shard_range(4000, r, 8) for r in range(8)][3] == (1500, 2000)


def test_bootstrap_exchange_times_out_without_a_publisher(pt, tmp_path):
    import pytest

    with pytest.raises(pt.PtError, match="timed out"):
        pt.bootstrap_exchange(str(tmp_path / "nobody"), 1, bytes(8), 0.2)


def test_two_rank_render_equals_single_process(orc, tmp_path):
    spp = 6
    out = str(tmp_path / "r.npz")
    mp.spawn(_worker, args=(2, _free_port(), spp, out, str(tmp_path / "rccl_id")), nprocs=2, join=True)
    got = np.load(out)
    s = orc.Scene()
    cam = s.build_scene(3, 24, spp)
    ref, cnt = s.render(cam, 1, 0, spp)
    assert int(got["spp"]) == spp and int(got["segments"]) == cnt["segments"]
    # disjoint sample ranges of the same (seed, pixel, sample)-keyed streams: identical sample
    # set, only the summation order differs
    np.testing.assert_allclose(got["accum"], ref, rtol=1e-13, atol=1e-13)
    s.close()


def _clean_env(tmp_path):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR", "PT_AMD_LAUNCH_TOKEN")}
    env["TMPDIR"] = str(tmp_path)
    return env


def test_bench_self_spawn_starts_its_ranks_and_relays_one_json_line(tmp_path):
    """`python bench.py --gpus 2` with no launcher: the parent must start two fresh rank processes (RANK / WORLD_SIZE /
    launch token set), relay exactly rank 0's JSON line and leave no rendezvous file behind. --rendezvous-only keeps the
    ranks to the host-side plumbing (pt_bootstrap_exchange through the C ABI): there is no GPU here; what runs behind the
    rendezvous on GPUs is covered by test_two_rank_render_equals_single_process (gloo / oracle standing in)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rendezvous-only"], env=_clean_env(tmp_path),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["launcher"] == "self" and d["rendezvous_only"] is True
    assert d["rank0_pid"] != os.getpid()
    assert os.path.dirname(d["id_path"]) == str(tmp_path)
    assert [f for f in os.listdir(tmp_path) if f.startswith("pt_amd_rccl_id_")] == []


def test_bench_under_a_launcher_is_one_rank_and_does_not_spawn(tmp_path):
    """With RANK in the environment (torchrun's contract) bench.py is ONE rank of the launch: it never starts children, and its
    rendezvous file is named by the launcher's run id, port and pid."""
    env = dict(_clean_env(tmp_path), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29411", TORCHELASTIC_RUN_ID="job7")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--rendezvous-only"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip())
    assert d["n_gpus"] == 1 and d["launcher"] == "external"          # the communicator's size, not the flag
    assert os.path.basename(d["id_path"]) == f"pt_amd_rccl_id_job7_29411_{os.getpid()}"
