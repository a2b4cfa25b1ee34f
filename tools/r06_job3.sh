#!/bin/bash
# the step with k_c3 in: conv tests + module tests on the GPU, bench line, A/B against MDS_KNOBS=22=1
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_k_conv.py tests/test_module_gpu.py -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r06j_tests.txt
for rep in 1 2; do
  python bench.py --no-pmc --no-other-configs --no-cpu-baseline --profile-steps 0 2>/dev/null | python tools/bench_brief.py | head -1 | sed 's/^/c3 on : /'
  MDS_KNOBS="22=1" python bench.py --no-pmc --no-other-configs --no-cpu-baseline --profile-steps 0 2>/dev/null | python tools/bench_brief.py | head -1 | sed 's/^/c3 off: /'
done 2>&1 | tee gpurun_out/r06j_ab.txt
python - <<'PY' 2>&1 | grep -v amdgpu | tee gpurun_out/r06j_mem.txt
import torch, time
d = torch.device("cuda:0")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
a = torch.empty(301_000_000 // 2, dtype=torch.bfloat16, device=d); b = torch.empty_like(a); c = torch.empty(75_000_000 // 2, dtype=torch.bfloat16, device=d)
us = t(lambda: a.fill_(1.0)); print(f"fill 301 MB: {us:.1f} us = {301e6 / us / 1e6:.2f} TB/s (pure write)")
us = t(lambda: b.copy_(a)); print(f"copy 301 MB: {us:.1f} us = {602e6 / us / 1e6:.2f} TB/s (read + write)")
us = t(lambda: c.sum()); print(f"sum 75 MB: {us:.1f} us")
PY
