#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_predictor.py tests/test_k_conv.py -m gpu -x -q -k "lane_selection or c3 or real_frame" 2>&1 | grep -E "passed|failed|error|Error|assert|Warning" | tail -6 | tee gpurun_out/r06aa_tests.txt
for rep in 1 2 3; do
  python bench.py --no-pmc --no-other-configs --no-cpu-baseline --profile-steps 0 2>/dev/null | python tools/bench_brief.py | head -1 | sed 's/^/HEAD (3 store waves for SILU post) : /'
  MDS_FUSE_CONV_POST_SILU=0 python bench.py --no-pmc --no-other-configs --no-cpu-baseline --profile-steps 0 2>/dev/null | python tools/bench_brief.py | head -1 | sed 's/^/post silu off                      : /'
done 2>&1 | tee gpurun_out/r06aa_ab.txt
