#!/bin/bash
mkdir -p gpurun_out
python tools/pwk_trace.py > gpurun_out/r05_pwk2_trace.txt 2>&1
cat gpurun_out/r05_pwk2_trace.txt
