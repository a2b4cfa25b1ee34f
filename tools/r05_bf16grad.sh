#!/bin/bash
mkdir -p gpurun_out
MDS_TEST_BF16_GRAD_BAR=1000 timeout 1500 python -m pytest tests/test_fullsize_gpu.py -q -m gpu -x -s -k "gradients_tensor_by_tensor" 2>&1 | grep -v "^$" | tail -12 | cut -c1-4000 > gpurun_out/r05_bf16_grad.txt
cat gpurun_out/r05_bf16_grad.txt
