#!/bin/bash
mkdir -p gpurun_out
( time python bench.py > gpurun_out/r05_bench_time.json 2> gpurun_out/r05_bench_time.err2 ) 2> gpurun_out/r05_bench_time.err
cat gpurun_out/r05_bench_time.err
python - <<'PY'
import json
d=json.loads([x for x in open('gpurun_out/r05_bench_time.json') if x.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['roofline'].get('traffic'), d['other_configs']['predict']['roofline'].get('traffic'), d['other_configs']['long004']['roofline'].get('kernel'))
PY
