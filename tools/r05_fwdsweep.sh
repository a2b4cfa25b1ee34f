#!/bin/bash
mkdir -p gpurun_out
for k in "" "19=64" "19=96" "19=128" "18=1" "20=1" ""; do MDS_KNOBS="$k" python tools/fwd_time.py 2>/dev/null | tail -1; done > gpurun_out/r05_fwd_knob_sweep4.txt
cat gpurun_out/r05_fwd_knob_sweep4.txt
