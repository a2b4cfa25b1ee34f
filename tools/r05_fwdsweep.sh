#!/bin/bash
mkdir -p gpurun_out
for k in "" "14=4" "14=8" "14=12" "14=16" "14=24" ""; do MDS_KNOBS="$k" python tools/fwd_time.py 2>/dev/null | tail -1; done > gpurun_out/r05_fwd_knob_sweep2.txt
cat gpurun_out/r05_fwd_knob_sweep2.txt
