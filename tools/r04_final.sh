#!/bin/bash
# round-4 profile set: GPU suite, default bench line (with other configs), kernel stats of the same command, PMC passes, step timeline
mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error|Error" | tail -5 > gpurun_out/r04_gpu_suite.txt
MDS_KEEP_TRACE_STATS=gpurun_out/r04_bench_kernel_stats.csv python bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err
SKIP_TRACE=1 bash tools/gpu_profile.sh r04p --no-other-configs > gpurun_out/r04_profile.log 2>&1
bash tools/r04_trace.sh
python tools/launch_table.py > gpurun_out/r04_launch_table.txt 2>&1
