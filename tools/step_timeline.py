"""Developer tool: one steady-state training step from a rocprofv3 --kernel-trace CSV, kernel by kernel on the dependent chain:
start offset, duration, gap since the chain's previous kernel ended, what the second stream ran meanwhile; totals per phase.
   python tools/step_timeline.py <kernel_trace.csv> [step index from the end, default 3]"""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3


def fam(n):
    n = n.replace("void ", "").split("<")[0].split("(")[0]
    return re.sub(r"_kernel$", "", n)


ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), fam(r["Kernel_Name"])) for r in rows))
opt = [k for k, e in enumerate(ev) if "adamw" in e[3]]
k1 = opt[-back]; k0 = opt[-back - 1]
step = ev[k0 + 1:k1 + 1]
qcount = collections.Counter(q for _, _, q, _ in step)
mainq = qcount.most_common(1)[0][0]
main = [e for e in step if e[2] == mainq]
side = [e for e in step if e[2] != mainq]
t0 = main[0][0]
print(f"step: {(ev[k1][1] - ev[k0][1]) / 1e3:.1f} us between optimizer ends; {len(main)} launches on the dependent chain, {len(side)} on the second stream")
first_side = side[0][0] if side else None
prev_end = None
tot = collections.defaultdict(float); gaps = collections.defaultdict(float)
for s, e, q, n in main:
    phase = "bwd" if first_side is not None and s >= first_side - 200000 and "focal" not in n else "fwd"
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    ov = [(sn, max(0, min(e, se) - max(s, ss)) / 1e3) for ss, se, _, sn in side if ss < e and se > s]
    tot[phase] += (e - s) / 1e3; gaps[phase] += max(gap, 0.0)
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} us  gap {gap:6.1f}  {n:22s} {'| ' + ', '.join(f'{a} {b:.0f}' for a, b in ov) if ov else ''}")
    prev_end = e
print("dependent chain: kernel time", {k: round(v) for k, v in tot.items()}, "us; gaps", {k: round(v) for k, v in gaps.items()}, "us")
if side:
    print(f"second stream: {sum(e - s for s, e, _, _ in side) / 1e3:.0f} us of kernels; last one ends {(max(e for _, e, _, _ in side) - main[-1][0]) / 1e3:.1f} us after the optimizer launch STARTS "
          f"(chain's last backward kernel ended {(main[-2][1] - main[-1][0]) / 1e3:.1f} us relative to it)")
