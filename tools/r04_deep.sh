#!/bin/bash
# fp32 inference pw_fwd with two K chunks in flight (DEEP): GPU parity + same-box A/B through the predictor bench (MDS_KNOBS=17=1: one chunk)
mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=gpurun_out/r04_ab_pw_deep.txt
python -m pytest tests/test_k_pw.py tests/test_predictor.py -q -m gpu 2>&1 | tail -3 > $OUT
echo "== predictor bench: knobs, chunks-of-8 fp32 frames/s, frame-by-frame fp32 / fp32 TTA / bf16" >> $OUT
for rep in 1 2; do
for k in "" "17=1" "10=2" "10=3" "10=1" "10=6"; do
  r=$(MDS_KNOBS="$k" python bench.py --config predict --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); f=d['frame_by_frame_api']; print(d['value'], f['fp32_frames_per_s'], f['fp32_tta_frames_per_s'], f['bf16_frames_per_s'], d['fp32_tta_frames_per_s'])")
  echo "KNOBS='$k' $r" >> $OUT
done; done
python tools/predict_timeline.py 1 > gpurun_out/r04_predict_timeline_deep.txt 2>&1
