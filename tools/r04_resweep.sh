#!/bin/bash
# switches / knobs decided in rounds 2-3, re-measured after round 4's reduction epilogues (same box, two repetitions)
OUT=gpurun_out/r04_resweep.txt; : > $OUT
run() { env "$@" python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-pmc --no-other-configs --profile-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
  for kv in X=1 MDS_FUSE_BN_BWD=0 MDS_STEM_DYP=0 MDS_KNOBS=8=1024 MDS_KNOBS=8=4096 MDS_KNOBS=12=1024 MDS_KNOBS=12=2560 MDS_KNOBS=13=200 MDS_KNOBS=13=800 MDS_KNOBS=15=400 MDS_KNOBS=15=1000 MDS_KNOBS=5=128 MDS_KNOBS=5=256; do
    echo "$kv $(run $kv)" >> $OUT
  done
done
cat $OUT
