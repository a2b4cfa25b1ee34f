cd ball-action-spotting_amd/csrc; cp libmds_hip.so libmds_new.so.bin
for v in new old; do cp libmds_$v.so.bin libmds_hip.so; echo $v; (cd ../..; python tools/probes/x3_error.py 368 640 2>&1 | tail -2); done
cp libmds_new.so.bin libmds_hip.so
