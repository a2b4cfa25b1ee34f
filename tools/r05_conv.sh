#!/bin/bash
mkdir -p gpurun_out
{ echo "## default prologue modes (as in the network), stats"; python tools/kbench.py conv_fwd 2>&1 | grep conv_fwd
echo "## no prologue, stats"; KB_MODE=0 python tools/kbench.py conv_fwd 2>&1 | grep conv_fwd
echo "## default prologue, no stats"; KB_NOSTATS=1 python tools/kbench.py conv_fwd 2>&1 | grep conv_fwd
echo "## no prologue, no stats"; KB_MODE=0 KB_NOSTATS=1 python tools/kbench.py conv_fwd 2>&1 | grep conv_fwd; } > gpurun_out/r05_conv_ablation.txt
cat gpurun_out/r05_conv_ablation.txt
