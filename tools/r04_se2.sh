#!/bin/bash
mkdir -p gpurun_out
for v in 0 1; do MDS_SE_FIN=$v python bench.py --profile-steps 2 --no-pmc --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); kb=d['kernel_breakdown']
print('SE_FIN=$v', d['ms_per_step'], {k: kb[k] for k in ('se_bwd_reduce','se_fc_bwd_data','bn_bwd_finalize','se_fc_bwd_params') if k in kb})" >> gpurun_out/r04_se2.txt; done
