#!/bin/bash
# Developer tool (GPU box): same-box A/B of whole-step time over (MDS_SIDE_CUS, MDS_KNOBS) pairs.
# usage: bash tools/ab_cus.sh out.txt "cus|knobs" ...      e.g. "0|" "128|" "128|6=4"
OUT=$1; shift
for rep in 1 2; do
for pair in "$@"; do
  cus=${pair%%|*}; k=${pair#*|}
  r=$(MDS_BENCH_STREAM="${BSTREAM:-0}" MDS_SIDE_PRIO="${PRIO:-}" MDS_SIDE_CUS="$cus" MDS_KNOBS="$k" timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-pmc --no-other-configs --profile-steps 0 2>gpurun_out/ab_cus_err.txt | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1)
  echo "cus='$cus' knobs='$k' $r" >> $OUT
done; done
