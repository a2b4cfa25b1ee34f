#!/bin/bash
# round-4 first GPU session: kernel tests of the fused apply + weight-gradient pass, isolated timings, in-step A/B
mkdir -p gpurun_out
python -m pytest tests/test_k_bwg.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r04_bwg_tests.txt
python tools/kbench.py bwg > gpurun_out/r04_bwg_kbench.txt 2>&1
KB_BWG_BLOCKS=256 python tools/kbench.py bwg > gpurun_out/r04_bwg_kbench_256.txt 2>&1
KB_BWG_BLOCKS=768 python tools/kbench.py bwg > gpurun_out/r04_bwg_kbench_768.txt 2>&1
bash tools/ab_env.sh gpurun_out/r04_ab_ride.txt MDS_WG_RIDE 0 1 2 3
python -m pytest tests/test_module_gpu.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r04_module_tests.txt
