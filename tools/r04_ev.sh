#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_module_gpu.py -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r04_ev_tests.txt
bash tools/ab_env.sh gpurun_out/r04_ab_events.txt MDS_SIDE_EVENTS record stop
MDS_WG_RIDE=3 bash tools/ab_env.sh gpurun_out/r04_ab_events_ride3.txt MDS_SIDE_EVENTS record stop
