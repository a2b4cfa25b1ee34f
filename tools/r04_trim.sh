#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/r04_trim.txt
python -m pytest tests/test_module_gpu.py tests/test_golden_hip.py tests/test_parallel.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error" | tail -3 >> gpurun_out/r04_trim.txt
for i in 1 2 3; do python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-pmc --no-other-configs --profile-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])" >> gpurun_out/r04_trim.txt; done
