#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/r04_steps.txt
for cfg in "20 5" "30 8" "40 10" "20 5" "60 15"; do set -- $cfg
  r=$(python bench.py --steps $1 --warmup $2 --no-cpu-baseline --no-pmc --no-other-configs --profile-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])")
  echo "steps $1 warmup $2: $r" >> gpurun_out/r04_steps.txt
done
