#!/bin/bash
# same-box A/B: blocks.0.0 weight gradient (behind the BN+SiLU prologue) through c3wp_kernel vs k_conv.hip
B="python bench.py --no-pmc --no-other-configs --no-cpu-baseline --profile-steps 0 --steps 150 --warmup 10"
for i in 1 2 3; do
  $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('c3wp blocks.0.0 wgrad', d['ms_per_step'], d['value'])"
  MDS_KNOBS="23=256" $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('k_conv (that layer)  ', d['ms_per_step'], d['value'])"
done
