#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error" | tail -5 > gpurun_out/r04m_gpu_suite.txt
bash tools/ab_lib.sh > gpurun_out/r04m_ab_dw3_occ.txt 2>&1     # libmds_old.so.bin = dw3_bwd at one block per CU
python tools/aug_bench.py 2>&1 | grep -v amdgpu > gpurun_out/r04m_aug_bench.txt
MDS_FOLD_BN=0 python bench.py --config predict --no-cpu-baseline > gpurun_out/r04m_predict_nofold.json 2>/dev/null
python bench.py --config predict --no-cpu-baseline --predict-kernel-trace > gpurun_out/r04m_predict.json 2>/dev/null
python bench.py --config long004 --profile-steps 0 --no-pmc --cpu-seconds 8 > gpurun_out/r04m_long004.json 2>/dev/null
