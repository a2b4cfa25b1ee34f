#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_k_dw_stem.py -q -m gpu -k stem -x 2>&1 | tail -2
python tools/launch_table.py 700 2>/dev/null | grep "stem_" 
run() { python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-pmc --no-other-configs --profile-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2 3; do echo "step $(run)"; done
