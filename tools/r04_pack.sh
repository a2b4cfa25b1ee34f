#!/bin/bash
# same-box A/B of the weight-pack kernel (LDS-tiled [O][I] -> [I][O] packs): new = libmds_hip.so, old = csrc/libmds_old.so.bin
mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=gpurun_out/r04_ab_pack.txt
python -m pytest tests/test_k_elem.py -q -m gpu -k pack 2>&1 | tail -2 > $OUT
echo "== step A/B (windows/s, ms per step)" >> $OUT
bash tools/ab_lib.sh >> $OUT 2>&1
C=ball-action-spotting_amd/csrc
for v in new old; do
  [ $v = old ] && cp $C/libmds_hip.so $C/libmds_new.so.bin && cp $C/libmds_old.so.bin $C/libmds_hip.so
  rm -rf /tmp/prof_pk && mkdir -p /tmp/prof_pk
  (cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_pk -o kt --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --profile-steps 0 --no-cpu-baseline --no-pmc --no-other-configs > /dev/null 2>&1)
  f=$(find /tmp/prof_pk -name '*kernel_stats.csv' | head -1)
  echo "== $v: pack kernel in the step (rocprofv3 --kernel-trace --stats: name, calls, total ns, avg ns)" >> $OUT
  grep -i "pack_kernel" $f | cut -d, -f1-4 >> $OUT
done
cp $C/libmds_new.so.bin $C/libmds_hip.so
