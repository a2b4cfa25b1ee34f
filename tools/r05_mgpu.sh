#!/bin/bash
# functional test of bench.py's N > 1 path on the one-GPU box (two ranks sharing device 0, gloo): NOT a scaling measurement
mkdir -p gpurun_out
MDS_BENCH_SHARED_GPU=1 timeout 900 python bench.py --gpus 2 --steps 6 --warmup 3 --batch 1 --no-cpu-baseline --no-pmc --no-other-configs > gpurun_out/r05_bench_shared_gpu_world2.json 2> gpurun_out/r05_bench_shared_gpu_world2.err
echo "rc=$?"; tail -3 gpurun_out/r05_bench_shared_gpu_world2.err | cut -c1-400
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r05_bench_shared_gpu_world2.json') if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print(d['value'], d['ms_per_step'], d['n_gpus']); print(json.dumps(d.get('parallel'))[:1500])
PY
