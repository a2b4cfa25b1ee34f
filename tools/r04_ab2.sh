#!/bin/bash
# generic same-box A/B: libmds_old.so.bin (HEAD) against the working-tree build; $1 = kbench targets (optional), $2 = repetitions
mkdir -p gpurun_out
cd ball-action-spotting_amd/csrc
cp libmds_hip.so libmds_new.so.bin
OUT=../../gpurun_out/r04_ab2.txt
: > $OUT
if [ -n "$1" ]; then
for v in old new; do
  cp libmds_$v.so.bin libmds_hip.so
  echo "== $v" >> $OUT
  (cd ../.. && python tools/kbench.py $1 2>&1 | grep -v "amdgpu.ids" >> gpurun_out/r04_ab2.txt)
done
fi
run() { (cd ../.. && python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-pmc --no-other-configs --profile-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"); }
for rep in $(seq 1 ${2:-3}); do
  cp libmds_old.so.bin libmds_hip.so; echo "old $(run)" >> $OUT
  cp libmds_new.so.bin libmds_hip.so; echo "new $(run)" >> $OUT
done
cp libmds_new.so.bin libmds_hip.so
cat $OUT
