#!/bin/bash
# same-box sweep at the end of round 6: the weight-gradient GEMMs' block budget (MDS_KNOB_WG_BLOCKS; 192 since round 5) now that the 3x3 layers own whole CUs
B="python bench.py --no-pmc --no-other-configs --no-cpu-baseline --profile-steps 0 --steps 120 --warmup 10"
for rep in 1 2; do
for v in 0 128 256 320 448; do
  MDS_KNOBS="5=$v" $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('wg_blocks $v  ', d['ms_per_step'], d['value'])"
done
done
