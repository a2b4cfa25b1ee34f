#!/bin/bash
# same-box A/B: the stride-2 forward layers (blocks.1.0 / 2.0) through k_c3.hip's c3s_kernel vs k_conv.hip
B="python bench.py --no-pmc --no-other-configs --no-cpu-baseline --profile-steps 0 --steps 150 --warmup 10"
for i in 1 2 3; do
  $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('c3s stride-2 forward ', d['ms_per_step'], d['value'])"
  MDS_KNOBS="23=64" $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('k_conv               ', d['ms_per_step'], d['value'])"
done
