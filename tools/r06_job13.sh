#!/bin/bash
# same-box A/B: the stride-1 3x3 weight gradients (blocks.1.1 / 2.1) through c3w_kernel vs k_conv.hip
B="python bench.py --no-pmc --no-other-configs --no-cpu-baseline --profile-steps 0 --steps 150 --warmup 10"
for i in 1 2 3; do
  $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('c3w weight gradient ', d['ms_per_step'], d['value'])"
  MDS_KNOBS="23=128" $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('k_conv wgrad        ', d['ms_per_step'], d['value'])"
done
