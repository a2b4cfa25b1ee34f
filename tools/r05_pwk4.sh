#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_k_pw.py -q -m gpu -k "kstream" -x 2>&1 | tail -3 > gpurun_out/r05_pwk8_tests.txt
cat gpurun_out/r05_pwk8_tests.txt
timeout 600 python tools/kbench.py pwk > gpurun_out/r05_pwk8_kbench.txt 2>&1
grep -v "^$" gpurun_out/r05_pwk8_kbench.txt | grep -v "b4\.\|b3\."
timeout 300 python tools/pwk_trace.py > gpurun_out/r05_pwk8_trace.txt 2>&1
bash tools/r05_ab1.sh
