"""Developer tool (no GPU needed): register spills of every kernel instantiation in csrc/*.hip, from hipcc's
-Rpass-analysis=kernel-resource-usage remarks.   python tools/spill_scan.py [min_spilled_vgprs]
Round 3 found the fp32 six-row dw2_fwd variant this way (56-63 spilled VGPRs at its three-blocks-per-CU bound: 2.2x slower)."""
import concurrent.futures, glob, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ball-action-spotting_amd", "csrc")
floor = int(sys.argv[1]) if len(sys.argv) > 1 else 1


def scan(src):
    err = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I.", "-Wno-unused-value", "-Wno-psabi",
                          "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"], cwd=CSRC, capture_output=True, text=True).stderr
    rows, cur = {}, None
    for l in err.splitlines():
        m = re.search(r"Function Name: (\S+)", l)
        if m:
            cur = m.group(1); rows[cur] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", l)
        if m and cur:
            rows[cur][m.group(1).strip()] = int(m.group(2))
    return os.path.basename(src), rows


with concurrent.futures.ThreadPoolExecutor(8) as ex:
    for name, rows in ex.map(scan, sorted(glob.glob(os.path.join(CSRC, "*.hip")))):
        for k, v in rows.items():
            if v.get("VGPRs Spill", 0) >= floor:
                dem = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip().split("(")[0]
                print(f"{name:12s} {dem[:90]:90s} vgpr {v.get('VGPRs'):4d} spilled {v.get('VGPRs Spill'):4d} scratch {v.get('ScratchSize'):5d} B  waves/SIMD {v.get('Occupancy')}")
