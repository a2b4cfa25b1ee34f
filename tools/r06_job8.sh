#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2; do
for k in 0 224 192 160 128; do
  MDS_KNOBS="24=$k" python bench.py --no-pmc --no-other-configs --no-cpu-baseline --profile-steps 0 2>/dev/null | python tools/bench_brief.py | head -1 | sed "s/^/c3 bwd blocks $k : /"
done; done 2>&1 | tee gpurun_out/r06y_blocks.txt
