#!/bin/bash
mkdir -p gpurun_out
cp ball-action-spotting_amd/csrc/libmds_trace.so.bin /tmp/base.bin
for v in base NOMMA SPLIT; do
[ $v = base ] && cp /tmp/base.bin ball-action-spotting_amd/csrc/libmds_trace.so.bin || cp ball-action-spotting_amd/csrc/libmds_trace_$v.so.bin ball-action-spotting_amd/csrc/libmds_trace.so.bin
echo "### $v"
timeout 300 python tools/pwk_trace.py 2>&1 | grep "==\|producer 0\|consumer 0" | head -15
done > gpurun_out/r05_pwk8_simd.txt
cat gpurun_out/r05_pwk8_simd.txt
