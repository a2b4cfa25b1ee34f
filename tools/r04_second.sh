#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_k_bwg.py -x -q -m gpu 2>&1 | tail -3 > gpurun_out/r04b_bwg_tests.txt
python tools/kbench.py bwg > gpurun_out/r04b_bwg_kbench.txt 2>&1
KB_BWG_BLOCKS=512 python tools/kbench.py bwg > gpurun_out/r04b_bwg_kbench_512.txt 2>&1
bash tools/ab_env.sh gpurun_out/r04b_ab_ride.txt MDS_WG_RIDE 0 1 2 3
