#!/bin/bash
# Developer tool (GPU box): same-box A/B of whole-step time over MDS_KNOBS settings.  usage: bash tools/ab_step.sh out.txt "4=32" "" "6=2" ...
OUT=$1; shift
for rep in 1 2; do
for k in "$@"; do
  r=$(MDS_KNOBS="$k" python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-pmc --no-other-configs --profile-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])")
  echo "KNOBS='$k' $r" >> $OUT
done; done
