"""hipcc -Rpass-analysis=kernel-resource-usage of one csrc/*.hip file as a table (developer tool).
   python tools/resusage.py k_pw.hip [filter]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "ball-action-spotting_amd", "csrc", sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-psabi",
                      "-c", src, "-o", "/tmp/resusage.o", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
cur = None
rows = {}
for l in out.splitlines():
    m = re.search(r"Function Name: (\S+)", l)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0]
        rows[cur] = {}
        continue
    m = re.search(r"remark: \s*([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", l)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
for k, v in rows.items():
    if flt in k:
        print(f"{k[:70]:70s} vgpr {v.get('VGPRs', 0):4d} agpr {v.get('AGPRs', 0):4d} spill {v.get('VGPR Spill', 0):3d} scratch {v.get('ScratchSize', 0):5d} occ {v.get('Occupancy', 0)} lds {v.get('LDS Size', 0)}")
