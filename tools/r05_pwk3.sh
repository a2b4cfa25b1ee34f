#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_k_pw.py -q -m gpu -k "kstream" -x 2>&1 | tail -5 > gpurun_out/r05_pwk3_tests.txt
python tools/kbench.py pwk > gpurun_out/r05_pwk3_kbench.txt 2>&1
python tools/pwk_trace.py > gpurun_out/r05_pwk3_trace.txt 2>&1
for k in "18=1" ""; do
  for rep in 1 2; do
    r=$(MDS_KNOBS="$k" python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-pmc --no-other-configs --profile-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])")
    echo "knobs=[$k] $r"
  done
done > gpurun_out/r05_pwk3_step.txt 2>&1
cat gpurun_out/r05_pwk3_tests.txt gpurun_out/r05_pwk3_step.txt gpurun_out/r05_pwk3_trace.txt
grep -v "^$" gpurun_out/r05_pwk3_kbench.txt
