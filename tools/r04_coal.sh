#!/bin/bash
# coalesced per-channel epilogues in the grouped / BatchNorm-backward reductions: isolated, then in-step A/B + knob sweeps
mkdir -p gpurun_out
cd ball-action-spotting_amd/csrc
cp libmds_hip.so libmds_new.so.bin
for v in old new; do
  cp libmds_$v.so.bin libmds_hip.so
  echo "== $v" >> ../../gpurun_out/r04_coal_kbench.txt
  (cd ../.. && python tools/kbench.py se 2>&1 | grep -v "amdgpu.ids" >> gpurun_out/r04_coal_kbench.txt)
done
run() { (cd ../.. && env MDS_KNOBS="$1" python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-pmc --no-other-configs --profile-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"); }
for rep in 1 2 3; do
  cp libmds_old.so.bin libmds_hip.so; echo "old $(run '')" >> ../../gpurun_out/r04_coal_ab.txt
  cp libmds_new.so.bin libmds_hip.so; echo "new $(run '')" >> ../../gpurun_out/r04_coal_ab.txt
done
cp libmds_new.so.bin libmds_hip.so
for rep in 1 2; do
for k in "14=16" "14=8" "14=64" "16=1024" "16=2048" "14=16,16=1024" ""; do
  echo "new knobs[$k] $(run "$k")" >> ../../gpurun_out/r04_coal_ab.txt
done; done
cat ../../gpurun_out/r04_coal_kbench.txt ../../gpurun_out/r04_coal_ab.txt
