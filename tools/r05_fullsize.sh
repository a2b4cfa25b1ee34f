#!/bin/bash
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_fullsize_gpu.py -q -m gpu -x -s -k "batch4 or full_window_vs_oracle" 2>&1 | grep -v "^$" | tail -25 > gpurun_out/r05_fullsize.txt
cat gpurun_out/r05_fullsize.txt
