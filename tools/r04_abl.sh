#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/r04_pwd_ablate.txt
for d in 0 1 2 4 8 16 3 5 12 15 31; do echo "== dbg $d" >> gpurun_out/r04_pwd_ablate.txt; KB_DBG=$d python tools/kbench.py pwd 2>&1 | grep "^pw_dgrad" >> gpurun_out/r04_pwd_ablate.txt; done
