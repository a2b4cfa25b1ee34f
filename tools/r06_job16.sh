#!/bin/bash
# same-box A/B at the end of round 6: BatchNorm-backward sums folded into the 3x3 data gradients (MDS_FUSE_CONV_POST=1) vs the default
B="python bench.py --no-pmc --no-other-configs --no-cpu-baseline --profile-steps 0 --steps 150 --warmup 10"
for i in 1 2 3; do
  $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('default             ', d['ms_per_step'], d['value'])"
  MDS_FUSE_CONV_POST=1 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('MDS_FUSE_CONV_POST=1', d['ms_per_step'], d['value'])"
done
