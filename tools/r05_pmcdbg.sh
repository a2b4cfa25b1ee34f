#!/bin/bash
( time python bench.py --config predict --steps 300 --predict-kernel-trace --no-cpu-baseline > gpurun_out/r05_predict_line.json 2>/dev/null ) 2>&1 | grep real
python - <<'PY'
import json
d=json.loads([x for x in open('gpurun_out/r05_predict_line.json') if x.startswith('{')][-1])
print(d['value'], d['frame_by_frame_api'], {k:d['roofline'].get(k) for k in ('kernel','frac','traffic','alg_bytes_per_launch','avg_launch_us')})
PY
