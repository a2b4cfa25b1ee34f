#!/bin/bash
# Developer tool (runs ON the GPU box through gpurun): bench line + rocprofv3 kernel stats + PMC passes.
#   gpurun --timeout 1500 -- 'bash tools/gpu_profile.sh r02a [extra bench args]'
# Writes gpurun_out/<tag>_*; the summaries worth judging are copied to profiles/ by hand afterwards.
# PMC passes are separate rocprofv3 runs with --pmc only (never combined with trace domains).
set -u
TAG=${1:-run}; shift || true
EXTRA="$*"
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
PY=python
BENCH="$ROOT/bench.py"
SHORT="--steps 3 --warmup 2 --profile-steps 0 --no-cpu-baseline --no-pmc --no-other-configs $EXTRA"   # (the child configs under --pmc took 40 GPU-minutes once)

echo "== bench" ; $PY $BENCH $EXTRA > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err ; tail -c 600 $OUT/${TAG}_bench.err

if [ "${SKIP_TRACE:-0}" != "1" ]; then
  echo "== kernel trace"
  rm -rf /tmp/prof_kt && mkdir -p /tmp/prof_kt
  (cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt --output-format csv -- $PY $BENCH --steps 10 --warmup 3 --profile-steps 0 --no-cpu-baseline --no-pmc --no-other-configs $EXTRA > $OUT/${TAG}_kt_bench.json 2> $OUT/${TAG}_kt.err)
  f=$(find /tmp/prof_kt -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/${TAG}_kernel_stats.csv
fi

pmc_pass() {  # name, counters...
  local name=$1; shift
  rm -rf /tmp/prof_$name && mkdir -p /tmp/prof_$name
  (cd /tmp && timeout 240 rocprofv3 --pmc "$@" -d /tmp/prof_$name -o pmc --output-format csv -- $PY $BENCH $SHORT > /dev/null 2> $OUT/${TAG}_pmc_$name.err)
  f=$(find /tmp/prof_$name -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && cp $f $OUT/${TAG}_pmc_$name.csv && echo "   $name: $(wc -l < $f) rows"
}
if [ "${SKIP_PMC:-0}" != "1" ]; then
  echo "== pmc passes"
  pmc_pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT
  pmc_pass grbm GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAIT_INST_LDS
  pmc_pass fetch FETCH_SIZE
  pmc_pass write WRITE_SIZE
  $PY $ROOT/tools/pmc_summary.py $OUT/${TAG}_pmc_hbm.json FETCH=$OUT/${TAG}_pmc_fetch.csv WRITE=$OUT/${TAG}_pmc_write.csv > $OUT/${TAG}_pmc_hbm.txt 2>&1
  $PY $ROOT/tools/pmc_mfma.py $OUT/${TAG}_pmc_mfma.json $OUT/${TAG}_pmc_sq.csv $OUT/${TAG}_pmc_grbm.csv > $OUT/${TAG}_pmc_mfma.txt 2>&1
fi
echo "== done"; ls -la $OUT | tail -20
