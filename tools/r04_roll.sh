#!/bin/bash
# rolling row slots in the streaming kernels: isolated (kbench se) per variant library, then the step A/B
mkdir -p gpurun_out
cd ball-action-spotting_amd/csrc
cp libmds_hip.so libmds_keep.so.bin
for v in old a b c d; do
  cp libmds_$v.so.bin libmds_hip.so
  echo "== $v" >> ../../gpurun_out/r04_roll_kbench.txt
  (cd ../.. && python tools/kbench.py se 2>&1 | grep -v "^$" >> gpurun_out/r04_roll_kbench.txt)
done
for rep in 1 2; do
for v in old a b c d; do
  cp libmds_$v.so.bin libmds_hip.so
  r=$(cd ../.. && python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-pmc --no-other-configs --profile-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])")
  echo "$v $r" >> ../../gpurun_out/r04_roll_ab.txt
done; done
cp libmds_keep.so.bin libmds_hip.so
cat ../../gpurun_out/r04_roll_kbench.txt ../../gpurun_out/r04_roll_ab.txt
