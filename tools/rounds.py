"""Developer tool (GPU box): how many ROUNDS of the chip every launch of one training step is.

A launch of B blocks on a kernel that fits k blocks per CU runs in ceil(B / (256 k)) rounds; a launch that is a few blocks past a
full round pays a whole extra one (the 3x3x3 depthwise kernels at 522 blocks for a capacity of 512 were the first case found).
Runs bench.py under `rocprofv3 --kernel-trace`, takes the last step's dispatches and prints grid, registers, LDS, blocks per CU,
rounds and duration, worst tail first.

  python tools/rounds.py [--config train] > gpurun_out/rounds.txt
"""
import csv, glob, math, os, shutil, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tmp = tempfile.mkdtemp(prefix="mds_rounds_", dir="/tmp")
steps = 3
cmd = ["rocprofv3", "--kernel-trace", "-d", tmp, "-o", "kt", "--output-format", "csv", "--", sys.executable, os.path.join(ROOT, "bench.py"),
       "--steps", str(steps), "--warmup", "2", "--profile-steps", "0", "--no-cpu-baseline", "--no-pmc", "--no-other-configs", *sys.argv[1:]]
subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
f = glob.glob(os.path.join(tmp, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows) // (steps + 2)
last = rows[-n:]
out = []
for r in last:
    wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    blocks = grid // wg
    vg = int(r.get("VGPR_Count", 0)) + int(r.get("Accum_VGPR_Count", 0))
    lds = int(r.get("LDS_Block_Size", 0))
    waves = max(wg // 64, 1)
    wps = min(8, 512 // max(vg, 64))                      # waves per SIMD by registers
    per_cu = max(1, (wps * 4) // waves)
    if lds:
        per_cu = max(1, min(per_cu, (160 * 1024) // lds))
    cap = 256 * per_cu
    rounds = blocks / cap
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:48]
    frac = rounds - math.floor(rounds)
    waste = (math.ceil(rounds) - rounds) / math.ceil(rounds) if rounds > 0 else 0
    out.append((us * waste if rounds > 1 else 0.0, us, name, blocks, wg, vg, lds, per_cu, rounds))
print(f"{'us':>8s} {'blocks':>7s} {'wg':>4s} {'vgpr':>4s} {'lds':>6s} {'b/CU':>4s} {'rounds':>7s}  kernel   (sorted by duration x idle share of the last round)")
for w, us, name, blocks, wg, vg, lds, per_cu, rounds in sorted(out, reverse=True)[:70]:
    print(f"{us:8.1f} {blocks:7d} {wg:4d} {vg:4d} {lds:6d} {per_cu:4d} {rounds:7.2f}  {name}")
print("\nfewer blocks than one round (latency class), by duration:")
for w, us, name, blocks, wg, vg, lds, per_cu, rounds in sorted([o for o in out if o[8] <= 1], key=lambda o: -o[1])[:50]:
    print(f"{us:8.1f} {blocks:7d} {wg:4d} {vg:4d} {lds:6d} {per_cu:4d} {rounds:7.2f}  {name}")
shutil.rmtree(tmp, ignore_errors=True)
