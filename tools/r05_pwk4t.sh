#!/bin/bash
mkdir -p gpurun_out
timeout 300 python tools/pwk_trace.py > gpurun_out/r05_pwk4_trace.txt 2>&1
cat gpurun_out/r05_pwk4_trace.txt
