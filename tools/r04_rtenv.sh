#!/bin/bash
# runtime switches of the HIP runtime that touch completion signals / stream waits (same-box A/B of the whole step)
OUT=gpurun_out/r04_ab_rtenv.txt; : > $OUT
run() { env "$@" python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-pmc --no-other-configs --profile-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
  echo "default $(run X=1)" >> $OUT
  echo "ROC_SYSTEM_SCOPE_SIGNAL=0 $(run ROC_SYSTEM_SCOPE_SIGNAL=0)" >> $OUT
  echo "GPU_STREAMOPS_CP_WAIT=1 $(run GPU_STREAMOPS_CP_WAIT=1)" >> $OUT
  echo "AMD_OPT_FLUSH=0 $(run AMD_OPT_FLUSH=0)" >> $OUT
  echo "ROC_ACTIVE_WAIT_TIMEOUT=100 $(run ROC_ACTIVE_WAIT_TIMEOUT=100)" >> $OUT
  echo "ROC_CPU_WAIT_FOR_SIGNAL=0 $(run ROC_CPU_WAIT_FOR_SIGNAL=0)" >> $OUT
  echo "DEBUG_HIP_DYNAMIC_QUEUES=0 $(run DEBUG_HIP_DYNAMIC_QUEUES=0)" >> $OUT
done
cat $OUT
