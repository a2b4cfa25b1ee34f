#!/bin/bash
mkdir -p gpurun_out
for pr in 0 1; do
cp ball-action-spotting_amd/csrc/libmds_trace_p$pr.so.bin ball-action-spotting_amd/csrc/libmds_trace.so.bin
echo "### consumer priority $pr (producers 3)"
timeout 300 python tools/pwk_trace.py 2>&1 | grep "==\|producer 0\|consumer 0" | head -15
done > gpurun_out/r05_pwk8_prio.txt
cat gpurun_out/r05_pwk8_prio.txt
