#!/usr/bin/env python3
"""bench.py — Msamples/s of the per-pixel integrator (Camera::render, camera.rs:79) on
BASELINE.json's headline config: scene 6 (OBJ meshes + envmap), 1920x1080 @ 4000 spp, f64.

One "step" = one complete pass of the hot path over one frame: every rank renders its contiguous
slice of the 4000 samples of every pixel (spp sharding, strong scaling: total work is fixed) into
its own device SUM accumulator, then ONE ncclReduce (RCCL over xGMI, called by libpt_amd.so itself:
pt_render_multi — no torch in the data path or anywhere else in this script) adds the accumulators
on rank 0. N=1 renders all 4000 spp on one GPU through the same entry point with a 1-rank
communicator and does no collective. The timed region is bracketed by a communicator barrier +
device synchronize on both sides (pt_comm_barrier) and the MAX over ranks of the elapsed time is
what the value is computed from.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--spp 4000] [--width 1920] [--scene 6]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Started plainly with --gpus N > 1 (no RANK in the environment) the script launches its N ranks ITSELF: the parent touches no
GPU, starts N fresh child processes of this file with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* and a launch token set,
relays rank 0's JSON line and exits with the worst child's code. Under a launcher (torchrun) it is one rank, as before.
`n_gpus` in the JSON is the communicator's size (pt_comm_world), never the flag.

Prints ONE JSON line on rank 0. `roofline` is measured live with HIP events on the launch stream
(pt_render's profile mode); `cpu_baseline` times the CPU oracle (kind "port": the faithful f64
restatement with the platform libm; the Rust reference itself cannot be built here) on a bounded
sample of the same workload, on rank 0 at N=1 only.
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time
import uuid

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0    # MI355X spec (MI355X_MICROARCH.md: 8.0 TB/s peak, ~6.3 achievable)
# algorithmic bytes: f64 path record = ray 7 f64 + throughput 3 f64 + (sample, draw, pixel) 3 u32 = 92 B
B_EXTEND_PER_SEGMENT = 56 + 4            # ray in (7 f64) + closest primitive out (u32)
B_SHADE_PER_SEGMENT = 92 + 4 + 92        # record in + closest primitive in + record out
B_FB_PER_SAMPLE = 24                     # 3 f64 accumulator add per finished sample


SCENE_NOTE = {3: "reference main.rs scene script: Cornell box, quad light, one-sample MIS",
              5: "reference main.rs scene script: 15 principled spheres, 7616x3808 environment map",
              6: "reference main.rs scene script: bunny+spot+cow OBJ meshes, envmap"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scene", type=int, default=6)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--spp", type=int, default=4000)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--slots-per-pixel", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--dump-frame", default=None, help="rank 0 saves the last timed frame's f64 sum accumulator (H, W, 3) as .npy (tests)")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="plumbing check without a GPU: every rank only runs the file rendezvous of pt_comm_create (pt_bootstrap_exchange) and rank 0 reports")
    return ap.parse_args()


def rendezvous_path():
    """The file through which rank 0 hands the RCCL unique id to the other ranks of THIS launch: named by a token every rank
    of the launch shares and no other launch has — PT_AMD_LAUNCH_TOKEN (self-spawn: a fresh uuid), else torchrun's run id +
    MASTER_PORT + the launcher's pid (the ranks of one torchrun are children of one agent process)."""
    token = os.environ.get("PT_AMD_LAUNCH_TOKEN")
    if not token:
        token = f"{os.environ.get('TORCHELASTIC_RUN_ID', 'none')}_{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}"
    return os.path.join(os.environ.get("TMPDIR", "/tmp"), f"pt_amd_rccl_id_{token}")


def spawn_ranks(n):
    """--gpus N without a launcher: start the N ranks as fresh child processes. Nothing in this (parent) process has
    touched the GPU or loaded the library."""
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    token = f"{uuid.uuid4().hex}_{os.getpid()}"
    base = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PT_AMD_LAUNCH_TOKEN=token,
                HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    stale = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"pt_amd_rccl_id_{token}")
    for f in (stale, stale + ".tmp"):             # (a fresh uuid: nothing can be there; removed before any child exists all the same)
        if os.path.exists(f):
            os.unlink(f)
    procs = []
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0, _ = procs[0].communicate()
    codes = [procs[0].returncode]
    for p in procs[1:]:
        try:
            codes.append(p.wait(timeout=60 if codes[0] == 0 else 5))
        except subprocess.TimeoutExpired:
            p.kill()                                  # our own child, by handle
            codes.append(p.wait())
    for f in (stale, stale + ".tmp"):
        try:
            os.unlink(f)
        except OSError:
            pass
    lines = [l for l in out0.decode(errors="replace").splitlines() if l.startswith("{")]
    if lines:
        print(lines[-1], flush=True)
    worst = max((abs(c) for c in codes), default=0)
    if worst == 0 and not lines:
        worst = 1
    sys.exit(worst)


def cpu_baseline(scene_id, width, seconds, images):
    """Times the oracle (the CHECKER, here only as the reported CPU baseline — never the product
    path) on all host cores: same scene, same resolution, as many spp as fit in ~`seconds`."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as orc

    orc.set_math_mode(False)                 # platform libm, like the Rust reference
    s = orc.Scene()
    cam = s.build_scene(scene_id, width, 1, images=images)     # BVH build is not timed (camera.rs:80 starts after it)
    h = orc.image_height(cam)
    s.render(cam, 1, 0, 1)                   # untimed: thread pool start-up, first touch of the scene
    t = time.time()
    s.render(cam, 1, 1, 3)
    per_spp = max((time.time() - t) / 2.0, 1e-3)
    spp = int(max(1, min(256, round(seconds / per_spp))))
    t = time.time()
    _, cnt = s.render(cam, 1, 3, 3 + spp)
    dt = time.time() - t
    s.close()
    cores = os.cpu_count()
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cpu_name = ""
    try:
        cpu_name = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        pass
    return {"value": round(width * h * spp / dt / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": f"scene {scene_id} {width}x{h} @ {spp} spp ({dt:.1f} s of CPU work, OpenMP over pixels, libm math)",
            "cpu": cpu_name, "segments_per_sample": round(cnt["segments"] / cnt["samples"], 4)}


def frame_check(pt, ctx, args, acc, height):
    """Sanity of the LAST timed frame (cheap, outside the timed region): finite, non-negative, right sample count and —
    for scenes the reference ships a render of — its 48x27 block means against demo/*.png (tests/golden)."""
    res = {"finite": bool(np.isfinite(acc).all()), "min": float(np.nanmin(acc)), "mean_linear": [round(float(x), 6) for x in acc.mean(axis=(0, 1)) / max(args.spp, 1)]}
    gold = os.path.join(ROOT, "tests", "golden", "reference_demo_blocks.npz")
    if os.path.exists(gold) and height % 27 == 0 and args.width % 48 == 0:
        ref = np.load(gold)
        key = f"scene{args.scene}"
        if key in ref.files:
            img = ctx.resolve_u8(acc, args.spp).astype(np.float64) / 255.0
            blocks = img.reshape(27, height // 27, 48, args.width // 48, 3).mean(axis=(1, 3))
            keep = np.ones((27, 48), dtype=bool)
            if args.scene == 6:
                keep[1:13, 27:36] = False        # the "spot" mesh of demo/scene6.png predates the mounted commit
            res["reference_demo_block_mad"] = [round(float(x), 5) for x in np.abs(blocks - ref[key])[keep].mean(axis=0)]
    return res


def main():
    args = parse()
    if "RANK" not in os.environ and args.gpus > 1:
        spawn_ranks(args.gpus)                        # does not return
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC between the ranks' processes (RCCL)

    pt = importlib.import_module("thu-acg-f2024-path-tracer_amd")
    id_path = rendezvous_path()
    if args.rendezvous_only:                           # host-side plumbing only (tests/test_distributed_cpu.py): no GPU, no render
        token = bytes(range(128)) if rank == 0 else bytes(128)
        got = pt.bootstrap_exchange(id_path, rank, token, 60.0) if world > 1 else token
        ok = got == bytes(range(128))
        if rank == 0:
            print(json.dumps({"rendezvous_only": True, "n_gpus": world, "rank0_pid": os.getpid(), "launcher": "self" if os.environ.get("PT_AMD_LAUNCH_TOKEN") else "external",
                              "id_path": id_path}), flush=True)
        sys.exit(0 if ok else 1)
    ctx = pt.Context(local_rank)                       # fails loudly without a GPU
    comm = pt.Comm(ctx, rank, world, id_path if world > 1 else None)
    world = comm.world                                 # the communicator's word, not the flag's or the environment's
    scene = pt.Scene(ctx)
    t0 = time.time()
    cam = scene.build_scene(args.scene, args.width, args.spp)
    build_s = time.time() - t0
    height = pt.image_height(cam)
    n_pix = args.width * height

    frame = np.zeros((height, args.width, 3), dtype=np.float64) if rank == 0 else None    # one landing buffer, overwritten by every frame

    def step(profile):
        return scene.render_multi(cam, args.seed, args.spp, comm, accum=frame, slots_per_pixel=args.slots_per_pixel, profile=profile, overwrite=True)

    for _ in range(args.warmup):
        step(False)
    comm.barrier()
    t_start = time.perf_counter()
    stats = []
    acc = None
    for _ in range(args.steps):
        acc, st = step(True)
        stats.append(st.as_dict())
    comm.barrier()
    elapsed = time.perf_counter() - t_start
    elapsed = float(comm.allreduce([elapsed], "max")[0])
    total_segments, total_samples = comm.allreduce([sum(s["segments"] for s in stats), sum(s["samples"] for s in stats)], "sum")
    if rank == 0 and world > 1:
        try:
            os.unlink(id_path)                         # every rank is past ncclCommInitRank
        except OSError:
            pass

    if rank == 0:
        samples_per_step = n_pix * args.spp
        value = samples_per_step * args.steps / elapsed / 1e6
        # dominant kernel on this rank, live HIP-event timing
        ms_ext = sum(s["ms_extend"] for s in stats); ms_sh = sum(s["ms_shade"] for s in stats)
        n_ext = sum(s["launches_extend"] for s in stats); n_sh = sum(s["launches_shade"] for s in stats)
        my_seg = sum(s["segments"] for s in stats); my_smp = sum(s["samples"] for s in stats)
        ext_name = {0: "k_extend2", 1: "k_extend"}.get(stats[0]["extend_variant"], "k_extend")
        sh_name = f"k_shade<{'true' if stats[0]['shade_variant'] >= 10 else 'false'}, {stats[0]['shade_variant'] % 10}>"
        if ms_ext >= ms_sh:
            kname, kms, kn = ext_name, ms_ext, n_ext
            bytes_total = my_seg * B_EXTEND_PER_SEGMENT
        else:
            kname, kms, kn = sh_name, ms_sh, n_sh
            bytes_total = my_seg * B_SHADE_PER_SEGMENT + my_smp * B_FB_PER_SAMPLE
        avg_ms = kms / max(kn, 1)
        achieved = (bytes_total / max(kn, 1)) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        # HBM bytes per launch from the PMC counters cannot be collected inside this process: they come from separate
        # `rocprofv3 --pmc` passes over this same command (tools/run_profiles.sh -> profiles/pmc_latest.json). The value is
        # attached only for the headline workload it was measured on, and labelled with its source; otherwise null.
        traffic, traffic_source = None, None
        pmc_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc_path) and (args.scene, args.width, args.spp, world) == (6, 1920, 4000, 1):
            try:
                pmc = json.load(open(pmc_path))
                if pmc.get("resident_paths") not in (None, stats[0]["n_slots"]):
                    raise ValueError("per-launch counters of another pool size")
                traffic = pmc.get("k_extend" if ms_ext >= ms_sh else "k_shade", {}).get("hbm_bytes_per_launch")
                traffic_source = f"profiles/pmc_latest.json ({pmc.get('source', 'earlier rocprofv3 --pmc passes of this command')}), not measured in this run"
            except Exception:
                traffic = None
        model = (f"algorithmic bytes = SURVEY §8(d) yardstick: K3 {B_SHADE_PER_SEGMENT} B/segment (92 B record in + 4 B hit word + 92 B record out) + {B_FB_PER_SAMPLE} B/sample "
                 f"(f64 accumulator), K2 {B_EXTEND_PER_SEGMENT} B/segment (56 B ray in + 4 B out); the code's compact records move less than the yardstick assumes (DESIGN.md §6)")
        roofline = {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "model": model,
                    "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_source": traffic_source, "kernel": kname,
                    "avg_launch_ms": round(avg_ms, 5), "launches": int(kn),
                    "algorithmic_bytes_per_launch": round(bytes_total / max(kn, 1), 1),
                    "other_kernel": {"name": sh_name if ms_ext >= ms_sh else ext_name,
                                     "avg_launch_ms": round((ms_sh if ms_ext >= ms_sh else ms_ext) / max(n_sh if ms_ext >= ms_sh else n_ext, 1), 5)},
                    "whole_pipeline_GBps": round((my_seg * (B_EXTEND_PER_SEGMENT + B_SHADE_PER_SEGMENT) + my_smp * B_FB_PER_SAMPLE) / max((ms_ext + ms_sh) * 1e-3, 1e-9) / 1e9, 3)}
        out = {
            "metric": "Msamples/s (WxHxspp/s), scene 6 FHD@4000spp" if (args.scene, args.width, args.spp) == (6, 1920, 4000) else "Msamples/s (WxHxspp/s)",
            "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"scene {args.scene} ({SCENE_NOTE.get(args.scene, 'reference main.rs scene script')}) {args.width}x{height} @ {args.spp} spp, max_depth 50, seed {args.seed}",
                       "parallelism": f"spp-sharded x{world}, one ncclReduce (RCCL, called by libpt_amd.so) of the f64 W*H*3 device accumulator" if world > 1 else "single GPU (1-rank communicator, no collective)",
                       "slots_per_pixel": stats[0]["slots_per_pixel"], "resident_paths": stats[0]["n_slots"],
                       "segments_per_sample": round(total_segments / max(total_samples, 1), 4), "scene_build_s": round(build_s, 3),
                       "device": ctx.name()},
            "roofline": roofline,
        }
        out["frame_check"] = frame_check(pt, ctx, args, acc, height)
        if args.dump_frame:
            np.save(args.dump_frame, acc)
        if world == 1 and not args.no_cpu_baseline:
            images = {n: pt.decode_image_rgb8(os.path.join(pt.ASSET_DIR, n)) for n in pt.SCENE_IMAGE_FILES.get(args.scene, [])}
            out["cpu_baseline"] = cpu_baseline(args.scene, args.width, args.cpu_seconds, images)
            out["gpu_over_cpu"] = round(value / max(out["cpu_baseline"]["value"], 1e-9), 1)
        print(json.dumps(out), flush=True)
    scene.close()
    comm.close()
    ctx.close()


if __name__ == "__main__":
    main()
