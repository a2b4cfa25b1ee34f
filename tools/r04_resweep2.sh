OUT=gpurun_out/r04_resweep2.txt; : > $OUT
run() { env "$@" python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-pmc --no-other-configs --profile-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
  for kv in X=1 MDS_KNOBS=2=1 MDS_KNOBS=2=3 MDS_KNOBS=14=16 MDS_KNOBS=14=24 MDS_KNOBS=14=48 MDS_KNOBS=1=1 MDS_KNOBS=1=2 MDS_KNOBS=9=16 MDS_KNOBS=9=48 MDS_KNOBS=7=8 MDS_KNOBS=7=12 MDS_KNOBS=0=768 MDS_KNOBS=0=1536 MDS_KNOBS=16=768 MDS_KNOBS=16=384; do
    echo "$kv $(run $kv)" >> $OUT
  done
done
cat $OUT
