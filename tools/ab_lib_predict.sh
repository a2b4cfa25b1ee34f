#!/bin/bash
# Developer tool (GPU box): A/B two builds of libmds_hip.so on the predictor bench (config 5), same box.
#   needs ball-action-spotting_amd/csrc/libmds_old.so.bin beside the current libmds_hip.so
cd ball-action-spotting_amd/csrc
cp libmds_hip.so libmds_new.so.bin
for rep in 1 2; do
for v in new old; do
  cp libmds_$v.so.bin libmds_hip.so
  (cd ../.. && python bench.py --config predict --no-cpu-baseline --no-pmc --no-other-configs --profile-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$v', 'chunk8 fp32', d['value'], 'tta', d.get('fp32_tta_frames_per_s'), 'bf16', d.get('bf16_frames_per_s'), 'frame by frame', json.dumps(d.get('frame_by_frame_api')))")
done; done
cp libmds_new.so.bin libmds_hip.so
