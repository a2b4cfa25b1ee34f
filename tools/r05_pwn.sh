#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_k_pw.py -q -m gpu -k "nstream or kstream" -x 2>&1 | tail -2
timeout 600 python tools/kbench.py pwn 2>&1 | grep -v "^$\|amdgpu.ids" > gpurun_out/r05_pwn_kbench.txt
cat gpurun_out/r05_pwn_kbench.txt

