#!/bin/bash
# round 6, first GPU job: bench line + in-step kernel stats at HEAD, the critical-path table, the GPU tests touched so far
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
MDS_KEEP_TRACE_STATS=gpurun_out/r06a python bench.py --no-pmc --trace-only --no-other-configs --no-cpu-baseline > gpurun_out/r06a_bench.json 2> gpurun_out/r06a_bench.err
python tools/bench_brief.py < gpurun_out/r06a_bench.json 2>/dev/null | head -5
python tools/critical_path.py --stats gpurun_out/r06a_bench_kernel_stats.csv > gpurun_out/r06a_critical_path.txt 2> gpurun_out/r06a_critical_path.err
tail -60 gpurun_out/r06a_critical_path.txt; tail -5 gpurun_out/r06a_critical_path.err
timeout 1500 python -m pytest tests/test_augment.py tests/test_predictor.py tests/test_train_ops.py -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/r06a_tests.txt
