"""Developer tool: how a training step ENDS (from a rocprofv3 --kernel-trace CSV): the last kernels of both queues before each
optimizer launch - is the weight-gradient stream's tail exposed?   python tools/step_tail.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), r["Kernel_Name"].replace("void ", "").split("<")[0].split("(")[0]) for r in rows))
opt = [k for k, e in enumerate(ev) if "adamw" in e[3] or "sgd_kernel" in e[3]]
for k in opt[len(opt) // 2: len(opt) // 2 + 3]:
    t_opt = ev[k][0]
    prev = opt[opt.index(k) - 1]
    step = ev[prev + 1:k]
    qs = {}
    for s, e, q, n in step:
        qs.setdefault(q, []).append((s, e, n))
    print(f"--- step of {(t_opt - ev[prev][1]) / 1e6:.2f} ms; optimizer starts at t = 0")
    for q, lst in qs.items():
        last = lst[-4:]
        busy = sum(e - s for s, e, _ in lst) / 1e6
        print(f" queue {q}: {len(lst)} launches, {busy:.2f} ms of kernels; last ones:", [(n[:18], round((s - t_opt) / 1e3, 1), round((e - t_opt) / 1e3, 1)) for s, e, n in last])
    # where is the side queue idle / busy over the step (ten slices)
    t0 = step[0][0]; L = (t_opt - t0) / 10
    for q, lst in qs.items():
        occ = [0.0] * 10
        for s, e, _ in lst:
            for b in range(10):
                lo, hi = t0 + b * L, t0 + (b + 1) * L
                occ[b] += max(0, min(e, hi) - max(s, lo)) / L
        print(f" queue {q} busy fraction per tenth of the step:", [round(o, 2) for o in occ])
