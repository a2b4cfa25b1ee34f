"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel name (developer tool).

  python tools/pmc_summary.py out.json FETCH=<dir_or_csv> WRITE=<dir_or_csv>
Applies the gfx950 corrections of MI355X_MICROARCH.md §HBM: FETCH_SIZE/WRITE_SIZE are in KiB and
FETCH_SIZE counts 128-byte requests as 64 bytes (x2)."""
import collections
import csv
import glob
import json
import os
import sys


def load(path, counter):
    files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
    tot, calls = collections.Counter(), collections.Counter()
    seen = set()
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"]
            tot[k] += float(r["Counter_Value"])
            key = (k, r.get("Dispatch_Id"))
            if key not in seen:
                seen.add(key)
                calls[k] += 1
    return tot, calls


if __name__ == "__main__":
    out = sys.argv[1]
    args = dict(a.split("=", 1) for a in sys.argv[2:])
    fetch, fc = load(args["FETCH"], "FETCH_SIZE")
    write, wc = load(args["WRITE"], "WRITE_SIZE")
    res = {}
    for k in fetch:
        n = max(fc[k], 1)
        res[k] = {"calls": n, "fetch_bytes_per_launch": fetch[k] * 1024 * 2 / n,
                  "write_bytes_per_launch": write.get(k, 0.0) * 1024 / max(wc.get(k, 1), 1)}
        res[k]["hbm_bytes_per_launch"] = res[k]["fetch_bytes_per_launch"] + res[k]["write_bytes_per_launch"]
    json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; KiB units; FETCH_SIZE x2 (gfx950)",
               "kernels": res}, open(out, "w"), indent=1)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["calls"])[:25]:
        print(f'{k[:60]:60s} n={v["calls"]:5d} fetch {v["fetch_bytes_per_launch"]/1e6:9.1f} MB  write {v["write_bytes_per_launch"]/1e6:9.1f} MB')
