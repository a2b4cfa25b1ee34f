#!/bin/bash
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > gpurun_out/gpu_suite.txt
cat gpurun_out/gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee -a gpurun_out/gpu_suite.txt
