#!/bin/bash
# k_c3.hip first contact: parity on the GPU, then the 3x3 layers alone with the new kernel on / off
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_k_conv.py -m gpu -x -q -k "c3" 2>&1 | tail -5 | tee gpurun_out/r06b_tests.txt
echo "== c3 on"; timeout 300 python tools/kbench.py conv_fwd conv_dgrad 2>&1 | tee gpurun_out/r06b_kbench_c3_on.txt
echo "== c3 off"; MDS_KNOBS="22=1" timeout 300 python tools/kbench.py conv_fwd conv_dgrad 2>&1 | tee gpurun_out/r06b_kbench_c3_off.txt
