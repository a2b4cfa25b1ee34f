#!/bin/bash
mkdir -p gpurun_out
python tools/launch_table.py 700 > gpurun_out/r05_base_launch_table.txt 2>&1
python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-pmc --no-other-configs --profile-steps 0 > gpurun_out/r05_base_bench.txt 2>&1
tail -2 gpurun_out/r05_base_bench.txt | cut -c1-600
