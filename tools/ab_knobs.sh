#!/bin/bash
# same-box A/B over MDS_KNOBS settings: bash tools/ab_knobs.sh out.txt "20=1" "" ...
OUT=$1; shift
mkdir -p gpurun_out
run() { python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-pmc --no-other-configs --profile-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2 3; do for k in "$@"; do echo "KNOBS='$k' $(MDS_KNOBS="$k" run)"; done; done > gpurun_out/$OUT
cat gpurun_out/$OUT
