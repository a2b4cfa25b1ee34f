#!/bin/bash
# in-step A/B: general kernel (knob 18=1) / K-streaming 8-wave kernel / the same with the materialised squeeze-excite activation
mkdir -p gpurun_out
run() { python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-pmc --no-other-configs --profile-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
  echo "general      $(MDS_KNOBS=18=1 run)"
  echo "kstream8     $(run)"
  echo "kstream8+act $(MDS_SE_ACT=1 run)"
  echo "general+act  $(MDS_KNOBS=18=1 MDS_SE_ACT=1 run)"
done > gpurun_out/r05_ab1.txt 2>&1
cat gpurun_out/r05_ab1.txt
