#!/bin/bash
# runtime switches not covered by tools/r04_rtenv.sh: kernel arguments in device memory, signal interrupts off
mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=gpurun_out/r04_rtenv2.txt
echo "== step A/B over runtime environment (windows/s, ms per step)" > $OUT
for rep in 1 2; do
for e in "X=0" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "HSA_ENABLE_INTERRUPT=0" "HIP_FORCE_DEV_KERNARG=1 HSA_ENABLE_INTERRUPT=0"; do
  r=$(env $e timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-pmc --no-other-configs --profile-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])")
  echo "$e : $r" >> $OUT
done; done
echo "== predictor, frame by frame fp32 / chunks of 8" >> $OUT
for e in "X=0" "HIP_FORCE_DEV_KERNARG=1" "HSA_ENABLE_INTERRUPT=0"; do
  r=$(env $e timeout 300 python bench.py --config predict --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['frame_by_frame_api']['fp32_frames_per_s'], d['value'])")
  echo "$e : $r" >> $OUT
done
