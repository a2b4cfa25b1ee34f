#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_module_gpu.py -x -q -m gpu 2>&1 | tail -30 > gpurun_out/r04_t_module.txt
MDS_PW_DGRAD=1 python -m pytest tests/test_module_gpu.py -x -q -m gpu 2>&1 | tail -40 > gpurun_out/r04_t_module_pwd.txt
