"""Per-launch durations of K2 / K3 over one frame, from a `rocprofv3 --kernel-trace --output-format csv` run:
how long the frame's END is (the sample budget is handed out, slots die, launches sweep a thinning pool).
Usage: python tools/trace_tail.py <dir-with-*kernel_trace.csv> [out.json]"""
import csv, glob, json, sys

d = sys.argv[1]
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
ext = [(e - s) / 1e6 for s, e, n in rows if "k_extend" in n]
sh = [(e - s) / 1e6 for s, e, n in rows if "k_shade" in n]
n = min(len(ext), len(sh))
it = [ext[i] + sh[i] for i in range(n)]
tot = sum(it)
full = sorted(it)[n // 2]                       # a typical full iteration
tail_start = next((i for i in range(n) if all(x < 0.9 * full for x in it[i:])), n)
res = {"launch_pairs": n, "sum_ms": round(tot, 2), "median_pair_ms": round(full, 4), "tail_starts_at": tail_start,
       "tail_pairs": n - tail_start, "tail_ms": round(sum(it[tail_start:]), 2), "tail_share": round(sum(it[tail_start:]) / tot, 4),
       "tail_if_full_ms": round((n - tail_start) * full, 2),
       "first_pairs_ms": [round(x, 3) for x in it[:6]], "every_16th_of_tail_ms": [round(x, 3) for x in it[tail_start::16]],
       "k2_ms": round(sum(ext), 2), "k3_ms": round(sum(sh), 2), "wall_first_to_last_ms": round((rows[-1][1] - rows[0][0]) / 1e6, 2) if rows else 0}
print(json.dumps(res))
if len(sys.argv) > 2:
    json.dump(res, open(sys.argv[2], "w"), indent=1)
