#!/bin/bash
# Developer tool (GPU box): same-box A/B of whole-step time over settings of ONE environment variable.
# usage: bash tools/ab_env.sh out.txt VAR v1 v2 ...
OUT=$1; VAR=$2; shift; shift
for rep in 1 2; do
for v in "$@"; do
  r=$(env $VAR="$v" python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-pmc --no-other-configs --profile-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])")
  echo "$VAR='$v' $r" >> $OUT
done; done
