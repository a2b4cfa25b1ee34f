#!/bin/bash
mkdir -p gpurun_out
run() { python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-pmc --no-other-configs --profile-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2 3; do
  echo "general       $(MDS_KNOBS=18=1 run)"
  echo "kstream8      $(run)"
  echo "kstream8 fwd  $(MDS_KNOBS=18=3 run)"
  echo "kstream8 dg   $(MDS_KNOBS=18=4 run)"
done > gpurun_out/r05_ab2.txt 2>&1
cat gpurun_out/r05_ab2.txt
