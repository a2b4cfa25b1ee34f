#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error" | tail -5 > gpurun_out/r04_gpu_suite.txt
bash tools/ab_env.sh gpurun_out/r04_ab_events2.txt MDS_SIDE_EVENTS record stop
