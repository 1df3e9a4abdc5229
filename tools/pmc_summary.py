"""Summarise rocprofv3 --pmc passes per kernel: mean per-launch value of every counter found under
<src>/*/ (one sub-directory per pass), HBM bytes per launch with the gfx950 corrections of
MI355X_MICROARCH.md §HBM (FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE under-reports wide coalesced reads by 2x —
also calibrated here on launches whose byte count is known), and the derived figures DESIGN.md §6 quotes.

    python tools/pmc_summary.py <src dir> <tag> [resident paths]  ->  profiles/<tag>_pmc_summary.json

Unit notes (MI355X_MICROARCH.md, cycle-constants row 's_memtime tick vs SQ PMC units'): SQ_WAVE_CYCLES, SQ_WAIT_* and
SQ_ACTIVE_INST_* count QUAD-cycles summed over waves; SQ_BUSY_CYCLES counts cycles per shader engine (32 SEs x 8 CUs
x 4 SIMDs). Mean resident waves per SIMD while busy = 4 * SQ_WAVE_CYCLES / SQ_BUSY_CYCLES / 32 (round 1 divided the
two raw numbers and reported an impossible 23.4: that is waves per SE in quad-cycle units)."""
import csv, glob, json, os, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, tag = sys.argv[1], sys.argv[2]
slots = int(sys.argv[3]) if len(sys.argv) > 3 else 33554432
KERNELS = ("k_extend2", "k_extend", "k_shade", "k_init", "k_resolve")


def short(name):
    for k in KERNELS:
        if k in name:
            return "k_extend" if k == "k_extend2" else k
    return None


vals = defaultdict(lambda: defaultdict(list))   # kernel -> counter -> [(dispatch, value summed over dims)]
for path in glob.glob(os.path.join(src, "**", "*_counter_collection.csv"), recursive=True):
    per, names = defaultdict(float), {}
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        if not k:
            continue
        per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
        names[r["Dispatch_Id"]] = k
    for (d, c), v in per.items():
        vals[names[d]][c].append((int(d), v))
out = {}
for k, cs in vals.items():
    o = {}
    for c, lst in cs.items():
        lst.sort()
        v = [x[1] for x in lst]
        o[c] = {"mean": sum(v) / len(v), "first": v[0], "max": max(v), "n": len(v)}
    out[k] = o


def m(k, c):
    return out[k][c]["mean"] if k in out and c in out[k] else None


cal = {}
if m("k_init", "WRITE_SIZE"):
    # k_init writes RayRec + hit_prim + state per slot (and a PathRec only without compact records: scenes with moving spheres)
    cal["write_factor"] = slots * (64 + 4 + 4) / (out["k_init"]["WRITE_SIZE"]["first"] * 1024)
if m("k_extend", "FETCH_SIZE"):
    cal["fetch_factor_first_extend"] = slots * (64 + 4) / (out["k_extend"]["FETCH_SIZE"]["first"] * 1024)   # first K2 launch: every slot alive
out["calibration"] = cal
out["resident_paths"] = slots   # the pool the per-launch figures belong to (bench.py attaches `traffic` only at the same pool size)
FF = 2.0   # guide: FETCH_SIZE reads exactly 1/2 of wide coalesced streams on gfx950 (own calibration above: 1.7-1.8)
for k in ("k_extend", "k_shade"):
    if k not in out:
        continue
    o, d = out[k], {}
    if m(k, "FETCH_SIZE") and m(k, "WRITE_SIZE"):
        o["hbm_fetch_bytes_per_launch"] = m(k, "FETCH_SIZE") * 1024 * FF
        o["hbm_write_bytes_per_launch"] = m(k, "WRITE_SIZE") * 1024
        o["hbm_bytes_per_launch"] = o["hbm_fetch_bytes_per_launch"] + o["hbm_write_bytes_per_launch"]
    wc = m(k, "SQ_WAVE_CYCLES")
    if wc:
        for name, c in (("wait_any_frac", "SQ_WAIT_ANY"), ("wait_inst_any_frac", "SQ_WAIT_INST_ANY"), ("wait_inst_lds_frac", "SQ_WAIT_INST_LDS"),
                        ("active_valu_frac", "SQ_ACTIVE_INST_VALU"), ("active_lds_frac", "SQ_ACTIVE_INST_LDS"), ("active_vmem_frac", "SQ_ACTIVE_INST_VMEM"),
                        ("active_scalar_frac", "SQ_ACTIVE_INST_SCA"), ("active_any_frac", "SQ_ACTIVE_INST_ANY")):
            if m(k, c) is not None:
                d[name + "_of_wave_cycles"] = m(k, c) / wc
        if m(k, "SQ_BUSY_CYCLES"):
            d["mean_waves_per_simd_while_busy"] = 4.0 * wc / m(k, "SQ_BUSY_CYCLES") / 32.0
    if m(k, "SQ_THREAD_CYCLES_VALU") and m(k, "SQ_ACTIVE_INST_VALU"):
        d["valu_thread_utilisation"] = m(k, "SQ_THREAD_CYCLES_VALU") / (m(k, "SQ_ACTIVE_INST_VALU") * 64)
    if m(k, "SQ_WAVES"):
        for c in ("SQ_INSTS_VALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_BRANCH"):
            if m(k, c) is not None:
                d[c.lower() + "_per_wave"] = m(k, c) / m(k, "SQ_WAVES")
    if m(k, "SQ_INST_LEVEL_VMEM") and m(k, "SQ_INSTS_VMEM_RD") is not None:
        n_vmem = m(k, "SQ_INSTS_VMEM_RD") + (m(k, "SQ_INSTS_VMEM_WR") or 0.0)
        d["mean_vmem_latency_cycles"] = m(k, "SQ_INST_LEVEL_VMEM") / max(n_vmem, 1.0)          # Little: sum of in-flight counts / instructions
    if m(k, "SQ_INST_LEVEL_LDS") and m(k, "SQ_INSTS_LDS"):
        d["mean_lds_latency_cycles"] = m(k, "SQ_INST_LEVEL_LDS") / m(k, "SQ_INSTS_LDS")
    if m(k, "SQ_INST_LEVEL_SMEM") and m(k, "SQ_INSTS_SMEM"):
        d["mean_smem_latency_cycles"] = m(k, "SQ_INST_LEVEL_SMEM") / m(k, "SQ_INSTS_SMEM")
    if m(k, "SQ_LDS_BANK_CONFLICT") is not None and m(k, "SQ_LDS_IDX_ACTIVE"):
        d["lds_bank_conflict_frac_of_lds_cycles"] = m(k, "SQ_LDS_BANK_CONFLICT") / m(k, "SQ_LDS_IDX_ACTIVE")
    if m(k, "TCC_HIT_sum") is not None and m(k, "TCC_MISS_sum") is not None:
        d["l2_hit_rate"] = m(k, "TCC_HIT_sum") / max(m(k, "TCC_HIT_sum") + m(k, "TCC_MISS_sum"), 1.0)
    if m(k, "TCP_TCC_READ_REQ_LATENCY_sum") and m(k, "TCP_TCC_READ_REQ_sum"):
        d["mean_l1_miss_read_latency_cycles"] = m(k, "TCP_TCC_READ_REQ_LATENCY_sum") / m(k, "TCP_TCC_READ_REQ_sum")
    if m(k, "TCP_TOTAL_CACHE_ACCESSES_sum") and m(k, "TCP_TCC_READ_REQ_sum") is not None:
        d["l1_read_requests_to_l2_per_cache_access"] = m(k, "TCP_TCC_READ_REQ_sum") / m(k, "TCP_TOTAL_CACHE_ACCESSES_sum")
    f64 = [m(k, c) for c in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64")]
    f32 = [m(k, c) for c in ("SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32")]
    if all(x is not None for x in f64) and m(k, "SQ_INSTS_VALU"):
        d["f64_arith_share_of_valu_insts"] = sum(f64) / m(k, "SQ_INSTS_VALU")
    if all(x is not None for x in f32) and m(k, "SQ_INSTS_VALU"):
        d["f32_arith_share_of_valu_insts"] = sum(f32) / m(k, "SQ_INSTS_VALU")
    if m(k, "SQ_INSTS_VALU_INT32") is not None and m(k, "SQ_INSTS_VALU"):
        d["int32_share_of_valu_insts"] = m(k, "SQ_INSTS_VALU_INT32") / m(k, "SQ_INSTS_VALU")
    o["derived"] = d
out["resident_paths"] = slots
out["source"] = (f"tools/run_profiles.sh {tag}: separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ_*, TCC_*; --kernel-trace only) over "
                 "`python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline`")
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_pmc_summary.json"), "w"), indent=1)
for k in ("k_extend", "k_shade"):
    if k in out:
        print(k, json.dumps({x: out[k][x] for x in out[k] if x in ("hbm_bytes_per_launch", "derived")}, indent=1))
print("calibration", cal)
