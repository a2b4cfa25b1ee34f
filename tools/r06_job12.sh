#!/bin/bash
# LDS counters of the row-streaming weight-gradient kernel alone (kbench conv_wgrad)
export TMPDIR=/tmp; rm -rf /tmp/prof_w; mkdir -p /tmp/prof_w gpurun_out
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d /tmp/prof_w -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/tools/kbench.py conv_wgrad > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/r06_c3w_pmc.err)
f=$(find /tmp/prof_w -name '*counter_collection.csv' | head -1)
python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:60]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    n[(k, r["Counter_Name"])] += 1
for k, d in acc.items():
    if "c3w" in k or "conv_wgrad" in k:
        print(k, {c: round(v / max(1, n[(k, c)])) for c, v in d.items()})
PY
