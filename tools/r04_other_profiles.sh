#!/bin/bash
# round-4 evidence for configs 4 and 5: rocprofv3 --kernel-trace --stats of the config-4 step and of the frame-by-frame predictor,
# the predictor's one-frame timeline
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf /tmp/prof_l4 && mkdir -p /tmp/prof_l4
(cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_l4 -o kt --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --config long004 --steps 10 --warmup 3 --profile-steps 0 --no-cpu-baseline --no-pmc --no-other-configs > $GRAFT_REPO_ROOT/gpurun_out/r04_long004_kt_bench.json 2> /dev/null)
f=$(find /tmp/prof_l4 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f gpurun_out/r04_long004_kernel_stats.csv
rm -rf /tmp/prof_pf && mkdir -p /tmp/prof_pf
(cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_pf -o kt --output-format csv -- python $GRAFT_REPO_ROOT/tools/predict_profile.py 200 1 > $GRAFT_REPO_ROOT/gpurun_out/r04_predict_fbf.log 2>&1)
f=$(find /tmp/prof_pf -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f gpurun_out/r04_predict_fbf_kernel_stats.csv
python tools/predict_timeline.py 1 > gpurun_out/r04_predict_timeline.txt 2>&1
python bench.py --config long004 --steps 20 --warmup 5 > gpurun_out/r04_bench_long004.json 2> /dev/null
python bench.py --config predict > gpurun_out/r04_bench_predict.json 2> /dev/null
