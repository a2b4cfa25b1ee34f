#!/bin/bash
# same-box A/B: squeeze-excite parameter gradients as one table launch per gradient bucket vs one launch per layer
python -m pytest tests/test_module_gpu.py tests/test_k_elem.py -m gpu -x -q 2>&1 | tail -2
B="python bench.py --no-pmc --no-other-configs --no-cpu-baseline --profile-steps 0 --steps 150 --warmup 10"
for i in 1 2 3; do
  $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('table per bucket   ', d['ms_per_step'], d['value'])"
  MDS_SE_PARAMS_TABLE=0 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('launch per layer   ', d['ms_per_step'], d['value'])"
done
