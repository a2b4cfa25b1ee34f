#!/bin/bash
# same-box A/B: the second-stream k_c3 kernels (c3w / c3wp) one priority level lower in every role (experiment library built by hand)
ARGS="--no-pmc --no-other-configs --no-cpu-baseline --profile-steps 0 --steps 150 --warmup 10"
run() { python - $ARGS <<PY 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', d['ms_per_step'], d['value'])"
import os, sys, runpy
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "ball-action-spotting_amd")]
from mds import cabi
if "$2": cabi.HIP_LIB = os.path.join(os.getcwd(), "ball-action-spotting_amd", "csrc", "$2")
sys.argv = ["bench.py"] + sys.argv[1:]
runpy.run_path("bench.py", run_name="__main__")
PY
}
for i in 1 2 3; do
  run "product priorities  " ""
  run "side kernels lower  " "libmds_lowprio.so.bin"
done
