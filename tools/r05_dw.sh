#!/bin/bash
mkdir -p gpurun_out
{ echo "## BN+SiLU prologue, stats"; python tools/kbench.py dw_fwd 2>&1 | grep dw_fwd
echo "## no prologue, stats"; KB_MODE=0 python tools/kbench.py dw_fwd 2>&1 | grep dw_fwd
echo "## BN+SiLU prologue, no stats"; KB_NOSTATS=1 python tools/kbench.py dw_fwd 2>&1 | grep dw_fwd
echo "## no prologue, no stats"; KB_MODE=0 KB_NOSTATS=1 python tools/kbench.py dw_fwd 2>&1 | grep dw_fwd
echo "## copy"; python tools/kbench.py copy 2>&1 | tail -4; } > gpurun_out/r05_dw_ablation.txt
cat gpurun_out/r05_dw_ablation.txt
