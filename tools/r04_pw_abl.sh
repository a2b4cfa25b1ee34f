cd ball-action-spotting_amd/csrc; cp libmds_hip.so keep.so.bin
for v in 0 1 2 3; do cp libmds_abl$v.so.bin libmds_hip.so; echo "== ablation $v (1: no atomics, 2: no cross-lane reduction, 3: neither)"; (cd ../.. && python tools/kbench.py pw_fwd 2>&1 | grep -v amdgpu.ids); done
cp keep.so.bin libmds_hip.so
