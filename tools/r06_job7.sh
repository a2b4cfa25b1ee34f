#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_k_conv.py tests/test_module_gpu.py tests/test_fullsize_gpu.py tests/test_golden_hip.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8 | tee gpurun_out/r06x_tests.txt
for rep in 1 2 3; do
  python bench.py --no-pmc --no-other-configs --no-cpu-baseline --profile-steps 0 2>/dev/null | python tools/bench_brief.py | head -1 | sed 's/^/HEAD          : /'
  MDS_FUSE_CONV_POST_SILU=0 python bench.py --no-pmc --no-other-configs --no-cpu-baseline --profile-steps 0 2>/dev/null | python tools/bench_brief.py | head -1 | sed 's/^/post silu off : /'
done 2>&1 | tee gpurun_out/r06x_ab.txt
