#!/bin/bash
mkdir -p gpurun_out
for k in 384 512 768 1024 1536 2048; do echo "## conv blocks $k"; MDS_KNOBS="0=$k" python tools/kbench.py conv_fwd 2>&1 | grep conv_fwd; done > gpurun_out/r05_conv_blocks_sweep.txt
cat gpurun_out/r05_conv_blocks_sweep.txt
