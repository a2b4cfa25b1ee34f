#!/bin/bash
# End-of-round evidence set (runs ON the GPU box through gpurun):  bash tools/round_profiles.sh r05
#   <tag>_bench_default.json                 the driver-style default `python bench.py` line (config 2 + other_configs)
#   <tag>_{bench,long004,predict_fbf}_kernel_stats.csv   rocprofv3 --kernel-trace --stats of each traced config (bench.py's own
#                                            child runs: MDS_KEEP_TRACE_STATS is a prefix, one file per config)
#   <tag>_pmc_hbm.{json,txt}, <tag>_pmc_mfma.{json,txt}   counter passes of the training step (tools/gpu_profile.sh)
#   <tag>_step_timeline.txt, <tag>_launch_table.txt, <tag>_predict_timeline.txt
TAG=${1:-rXX}
mkdir -p gpurun_out; export TMPDIR=/tmp
MDS_KEEP_TRACE_STATS=gpurun_out/${TAG} python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err
SKIP_TRACE=1 bash tools/gpu_profile.sh ${TAG}p --no-other-configs > gpurun_out/${TAG}_profile.log 2>&1
for k in hbm mfma; do for e in json txt; do [ -f gpurun_out/${TAG}p_pmc_$k.$e ] && cp gpurun_out/${TAG}p_pmc_$k.$e gpurun_out/${TAG}_pmc_$k.$e; done; done
rm -rf /tmp/prof_kt && mkdir -p /tmp/prof_kt
(cd /tmp && rocprofv3 --kernel-trace -d /tmp/prof_kt -o kt --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 3 --profile-steps 0 --no-cpu-baseline --no-pmc --no-other-configs > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/${TAG}_trace.err)
f=$(find /tmp/prof_kt -name '*kernel_trace.csv' | head -1)
python tools/step_timeline.py $f 3 > gpurun_out/${TAG}_step_timeline.txt 2>&1
python tools/launch_table.py 700 > gpurun_out/${TAG}_launch_table.txt 2>&1
python tools/predict_timeline.py 1 > gpurun_out/${TAG}_predict_timeline.txt 2>&1
python tools/bench_brief.py < gpurun_out/${TAG}_bench_default.json 2>/dev/null | head -12
ls -la gpurun_out/${TAG}_*
