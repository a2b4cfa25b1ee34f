"""Idle-time analysis of a rocprofv3 --kernel-trace CSV: per-stream busy time, gaps between consecutive
kernels, and the kernels that precede the largest gaps (developer tool)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# keep the last `frac` of the trace (steady state)
t0 = int(rows[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in rows)
lo = t0 + (t1 - t0) * float(sys.argv[2]) if len(sys.argv) > 2 else t0
rows = [r for r in rows if int(r["Start_Timestamp"]) >= lo]
span = (max(int(r["End_Timestamp"]) for r in rows) - int(rows[0]["Start_Timestamp"])) / 1e6
busy = 0; cur_end = 0; gaps = collections.Counter(); gapn = collections.Counter(); tot_gap = 0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if cur_end and s > cur_end:
        g = (s - cur_end) / 1e3
        tot_gap += g
        gaps[prev] += g; gapn[prev] += 1
    if e > cur_end:
        busy += (e - max(s, cur_end)) / 1e6
        cur_end = e; prev = r["Kernel_Name"][:40]
print(f"span {span:.2f} ms, union-busy {busy:.2f} ms, idle {tot_gap/1e3:.2f} ms over {len(rows)} kernels")
for k, v in gaps.most_common(15):
    print(f"  after {k:40s} idle {v/1e3:7.3f} ms  n={gapn[k]}  avg {v/gapn[k]:.1f} us")
