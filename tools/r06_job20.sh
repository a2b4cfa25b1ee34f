#!/bin/bash
# same-box A/B: blocks.2.0 weight gradient (stride 2) through c3w2_kernel vs k_conv.hip
B="python bench.py --no-pmc --no-other-configs --no-cpu-baseline --profile-steps 0 --steps 150 --warmup 10"
for i in 1 2 3; do
  $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('c3w2 blocks.2.0 wgrad', d['ms_per_step'], d['value'])"
  MDS_KNOBS="23=512" $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('k_conv (that layer) ', d['ms_per_step'], d['value'])"
done
