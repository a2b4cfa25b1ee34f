#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_k_pwd.py tests/test_k_bwg.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error" > gpurun_out/r04_pwd_tests.txt
python tools/kbench.py pwd 2>&1 | grep -v amdgpu > gpurun_out/r04_pwd_kbench.txt
bash tools/ab_env.sh gpurun_out/r04_ab_pwd.txt MDS_PW_DGRAD 0 1
python -m pytest tests/test_module_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error" > gpurun_out/r04_pwd_module_tests.txt
