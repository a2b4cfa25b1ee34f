#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
MDS_KEEP_TRACE_STATS=gpurun_out/r05b python bench.py > gpurun_out/r05b_bench_default.json 2> gpurun_out/r05b_bench_default.err
python tools/bench_brief.py < gpurun_out/r05b_bench_default.json 2>/dev/null | head -3 | cut -c1-400
ls gpurun_out/r05b_*
