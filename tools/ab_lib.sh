#!/bin/bash
# A/B two builds of libmds_hip.so on the same box
cd ball-action-spotting_amd/csrc
cp libmds_hip.so libmds_new.so.bin
for rep in 1 2 3; do
for v in new old; do
  cp libmds_$v.so.bin libmds_hip.so
  r=$(cd ../.. && python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-pmc --no-other-configs --profile-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])")
  echo "$v $r"
done; done
cp libmds_new.so.bin libmds_hip.so
