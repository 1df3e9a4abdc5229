"""Summarise rocprofv3 --pmc passes (tools/run_pmc.sh) per kernel: mean per-launch counter values,
HBM bytes per launch with the gfx950 corrections of MI355X_MICROARCH.md §HBM (FETCH_SIZE/WRITE_SIZE
are in KiB; FETCH_SIZE under-reports wide coalesced reads by 2x — calibrated here on launches whose
byte count is known), occupancy and VALU utilisation. Writes profiles/<tag>_pmc_summary.json and
profiles/pmc_latest.json (read by bench.py for roofline.traffic)."""
import csv, glob, json, os, sys
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]; tag = sys.argv[2]
def short(name):
    for k in ("k_extend_tlas", "k_extend_mesh", "k_extend", "k_shade", "k_init", "k_resolve", "k_compact"):
        if k in name: return k
    return None
vals = defaultdict(lambda: defaultdict(list))   # kernel -> counter -> per-dispatch values (summed over dims)
for path in glob.glob(os.path.join(src, "*", "*", "*_counter_collection.csv")):
    per = defaultdict(float); names = {}
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        if not k: continue
        key = (r["Dispatch_Id"], r["Counter_Name"]); per[key] += float(r["Counter_Value"]); names[r["Dispatch_Id"]] = k
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
# calibration of the byte counters on launches with known traffic (dynamic mode, 4,194,304 slots)
slots = int(sys.argv[3]) if len(sys.argv) > 3 else 16777216   # resident paths of the profiled run (256 CUs x 65536)
cal = {}
if "k_init" in out and "WRITE_SIZE" in out["k_init"]:
    known = slots * (64 + 32 + 4 + 4)                     # k_init (dynamic mode) writes RayRec + PathRec + hit_prim + bounce per slot
    cal["write_factor"] = known / (out["k_init"]["WRITE_SIZE"]["first"] * 1024)
if "k_extend" in out and "FETCH_SIZE" in out["k_extend"]:
    known = slots * (64 + 4)                              # first k_extend launch: every slot alive, RayRec + bounce
    cal["fetch_factor_first_extend"] = known / (out["k_extend"]["FETCH_SIZE"]["first"] * 1024)
out["calibration"] = cal
ff = 2.0    # guide: FETCH_SIZE reads exactly 1/2 of wide coalesced streams on gfx950
for k in ("k_extend", "k_extend_tlas", "k_extend_mesh", "k_shade"):
    if k in out and "FETCH_SIZE" in out[k] and "WRITE_SIZE" in out[k]:
        f = out[k]["FETCH_SIZE"]["mean"] * 1024 * ff; w = out[k]["WRITE_SIZE"]["mean"] * 1024
        out[k]["hbm_bytes_per_launch"] = f + w
        out[k]["hbm_fetch_bytes_per_launch"] = f; out[k]["hbm_write_bytes_per_launch"] = w
    if k in out and "SQ_WAVE_CYCLES" in out[k]:
        o = out[k]
        o["derived"] = {
            "valu_insts_per_wave": o["SQ_INSTS_VALU"]["mean"] / max(o["SQ_WAVES"]["mean"], 1),
            "valu_thread_utilisation": o["SQ_THREAD_CYCLES_VALU"]["mean"] / max(o["SQ_ACTIVE_INST_VALU"]["mean"] * 64, 1),
            "wait_inst_any_frac_of_wave_cycles": o["SQ_WAIT_INST_ANY"]["mean"] / max(o["SQ_WAVE_CYCLES"]["mean"], 1),
            "wait_any_frac_of_wave_cycles": o["SQ_WAIT_ANY"]["mean"] / max(o["SQ_WAVE_CYCLES"]["mean"], 1),
            "active_valu_frac_of_wave_cycles": o["SQ_ACTIVE_INST_VALU"]["mean"] / max(o["SQ_WAVE_CYCLES"]["mean"], 1),
            "mean_waves_per_simd_while_busy": o["SQ_WAVE_CYCLES"]["mean"] / max(o["SQ_BUSY_CYCLES"]["mean"], 1),
        }
    if k in out and "TCC_HIT_sum" in out[k]:
        h, m = out[k]["TCC_HIT_sum"]["mean"], out[k]["TCC_MISS_sum"]["mean"]
        out[k]["l2_hit_rate"] = h / max(h + m, 1)
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
for name in (f"{tag}_pmc_summary.json", "pmc_latest.json"):
    json.dump(out, open(os.path.join(ROOT, "profiles", name), "w"), indent=1)
for k in ("k_extend", "k_extend_tlas", "k_extend_mesh", "k_shade"):
    if k in out:
        print(k, json.dumps({x: out[k][x] for x in out[k] if x in ("hbm_bytes_per_launch", "derived", "l2_hit_rate")}, indent=1))
print("calibration", cal)
