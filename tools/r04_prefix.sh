#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_module_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error" | tail -3 > gpurun_out/r04_prefix_tests.txt
bash tools/ab_env.sh gpurun_out/r04_ab_prefix.txt MDS_PREFIX_SIDE 0 1
