"""GPU timeline of the training step from a rocprofv3 --kernel-trace CSV (developer tool): busy union, per-queue busy time,
idle gaps (is the host keeping up?), overlap of the weight-gradient stream.
   python tools/timeline.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
t_lo = ev[int(len(ev) * 0.4)][0]          # steady state: the last 60 % of the dispatches
ev = [e for e in ev if e[0] >= t_lo]
span = ev[-1][1] - ev[0][0]
busy = 0; cur_s, cur_e = ev[0][0], ev[0][1]; gaps = []
for s, e, q, n in ev[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append((s - cur_e, n)); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
perq = collections.defaultdict(int)
for s, e, q, n in ev:
    perq[q] += e - s
print(f"span {span / 1e6:.2f} ms, busy union {busy / 1e6:.2f} ms ({100 * busy / span:.1f} %), idle {100 - 100 * busy / span:.1f} %, {len(ev)} dispatches")
for q, t in sorted(perq.items(), key=lambda kv: -kv[1]):
    print(f"  queue {q}: kernel time {t / 1e6:.2f} ms ({100 * t / span:.1f} % of the span)")
gaps.sort(reverse=True)
print("largest idle gaps (us) and the kernel that ended them:")
for g, n in gaps[:10]:
    print(f"  {g / 1e3:8.1f}  {n[:70]}")
hist = collections.Counter(min(int(g / 1e3), 20) for g, _ in gaps)
print("gap histogram {us: (count, total us)}:", {k: (hist[k], round(sum(g for g, _ in gaps if min(int(g / 1e3), 20) == k) / 1e3)) for k in sorted(hist)})
