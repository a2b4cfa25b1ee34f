#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf /tmp/prof_kt && mkdir -p /tmp/prof_kt
(cd /tmp && rocprofv3 --kernel-trace -d /tmp/prof_kt -o kt --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 3 --profile-steps 0 --no-cpu-baseline --no-pmc --no-other-configs > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/r04_trace.err)
f=$(find /tmp/prof_kt -name '*kernel_trace.csv' | head -1)
python tools/step_timeline.py $f 3 > gpurun_out/r04_step_timeline.txt 2>&1
python tools/timeline.py $f > gpurun_out/r04_timeline.txt 2>&1
