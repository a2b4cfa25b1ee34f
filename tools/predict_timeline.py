"""Developer tool (GPU box): every launch of ONE steady-state frame of the frame-by-frame StreamPredictor, in order:
name, blocks, duration, gap to the previous launch.   python tools/predict_timeline.py [chunk] > gpurun_out/predict_timeline.txt"""
import csv, glob, os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
chunk = sys.argv[1] if len(sys.argv) > 1 else "1"
tmp = tempfile.mkdtemp(prefix="mds_ptl_", dir="/tmp")
n = 40 * int(chunk)
subprocess.run(["rocprofv3", "--kernel-trace", "-d", tmp, "-o", "kt", "--output-format", "csv", "--", sys.executable,
                os.path.join(ROOT, "tools", "predict_profile.py"), str(n), chunk], cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"),
               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900)
f = glob.glob(os.path.join(tmp, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# one frame = from one stem_fwd launch to the next; take the last complete one
stems = [i for i, r in enumerate(rows) if "stem_fwd" in r["Kernel_Name"]]
a, b = stems[-2], stems[-1]
prev_end = int(rows[a - 1]["End_Timestamp"]) if a else int(rows[a]["Start_Timestamp"])
tot = gaps = 0.0
print(f"{'us':>8s} {'gap':>6s} {'blocks':>7s}  kernel")
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    blocks = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // wg
    us, gap = (e - s) * 1e-3, (s - prev_end) * 1e-3
    tot += us; gaps += max(gap, 0.0)
    print(f"{us:8.1f} {gap:6.1f} {blocks:7d}  {r['Kernel_Name'].replace('void ', '')[:90]}")
    prev_end = e
print(f"launches {b - a}, kernel time {tot:.1f} us, gaps {gaps:.1f} us, frame {(int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp'])) * 1e-3:.1f} us")
shutil.rmtree(tmp, ignore_errors=True)
