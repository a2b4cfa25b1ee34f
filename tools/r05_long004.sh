#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_k_dw_stem.py -q -m gpu -k "dw_fwd_bwd or strip_lengths" 2>&1 | tail -1
python tools/kbench.py dw_bwd 2>&1 | grep "3d"
MDS_KNOBS=21=2 python tools/kbench.py dw_bwd 2>&1 | grep "3d 11"
for L in 5 8 13 20 40; do MDS_KNOBS=7=$L python tools/kbench.py dw_bwd 2>&1 | grep "3d 11"; done
run() { python bench.py --config long004 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-other-configs --profile-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2 3; do echo "LDS-tiled dw3 fwd+bwd (21=1) $(MDS_KNOBS=21=1 run)"; echo "dw3g fwd only (21=2)         $(MDS_KNOBS=21=2 run)"; echo "dw3g fwd + bwd               $(run)"; done | tee gpurun_out/r05_ab_long004_dw3g.txt
