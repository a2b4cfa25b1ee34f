"""Summarise rocprofv3 --pmc SQ/GRBM passes per kernel (developer tool).

  python tools/pmc_mfma.py out.json sq_pass.csv [grbm_pass.csv ...]

Per kernel name: launches, mean of every collected counter per launch, and the ratios DESIGN.md quotes
  wave_parked   = SQ_WAIT_ANY / SQ_WAVE_CYCLES          (s_waitcnt / barrier)
  issue_stall   = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES
  issuing       = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES
  mfma_busy     = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel duration x 2.4 GHz)   [upper clock; DVFS
                  makes the true fraction higher] and, when GRBM_GUI_ACTIVE is present, / (GUI_ACTIVE x 1024 / xcc)
  lds_conflict  = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (when both were collected)
Counters are summed over all instances by rocprofv3; durations come from the dispatch timestamps."""
import collections
import csv
import json
import re
import sys

SIMDS, CLOCK = 1024, 2.4e9


def short(name):
    n = name.replace("void ", "")
    n = re.sub(r"\(.*$", "", n)
    return n


def main():
    out = sys.argv[1]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.defaultdict(int))
    dur = collections.defaultdict(float)
    ndur = collections.defaultdict(int)
    for f in sys.argv[2:]:
        seen = set()
        try:
            rows = list(csv.DictReader(open(f)))
        except OSError:
            continue
        for r in rows:
            k = short(r["Kernel_Name"])
            c = r["Counter_Name"]
            agg[k][c] += float(r["Counter_Value"])
            key = (k, r.get("Dispatch_Id"), c)
            if key not in seen:
                seen.add(key)
                cnt[k][c] += 1
            dk = (f, k, r.get("Dispatch_Id"))
            if dk not in seen and r.get("Start_Timestamp") and r.get("End_Timestamp"):
                seen.add(dk)
                dur[k] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-9
                ndur[k] += 1
    res = {}
    for k, d in agg.items():
        e = {"launches": max(cnt[k].values())}
        for c, v in d.items():
            e[c] = v / max(cnt[k][c], 1)
        wc = e.get("SQ_WAVE_CYCLES")
        if wc:
            for nm, c in (("wave_parked", "SQ_WAIT_ANY"), ("issue_stall", "SQ_WAIT_INST_ANY"), ("issuing", "SQ_ACTIVE_INST_ANY"),
                          ("valu_issuing", "SQ_ACTIVE_INST_VALU")):
                if c in e:
                    e[nm] = round(e[c] / wc, 4)
        if ndur[k]:
            e["avg_us_profiled"] = round(dur[k] / ndur[k] * 1e6, 2)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in e:
                e["mfma_busy_at_2.4GHz"] = round(e["SQ_VALU_MFMA_BUSY_CYCLES"] / (SIMDS * (dur[k] / ndur[k]) * CLOCK), 4)
        if "SQ_LDS_BANK_CONFLICT" in e and e.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_conflict"] = round(e["SQ_LDS_BANK_CONFLICT"] / e["SQ_LDS_IDX_ACTIVE"], 4)
        res[k] = e
    json.dump({"note": __doc__.split("\n\n")[1] if "\n\n" in __doc__ else "", "kernels": res}, open(out, "w"), indent=1)
    tot = sorted(res.items(), key=lambda kv: -kv[1].get("avg_us_profiled", 0) * kv[1]["launches"])
    for k, e in tot[:30]:
        print(f'{k[:56]:56s} n={e["launches"]:5d} us={e.get("avg_us_profiled", 0):8.1f} mfma_busy={e.get("mfma_busy_at_2.4GHz", 0):.3f} '
              f'parked={e.get("wave_parked", 0):.2f} stall={e.get("issue_stall", 0):.2f} issuing={e.get("issuing", 0):.2f}')


if __name__ == "__main__":
    main()
