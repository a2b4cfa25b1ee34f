python -m pytest tests/test_k_pw.py tests/test_k_dw_stem.py tests/test_predictor.py tests/test_abi.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|assert|^E " | head -20
for kn in "10=1,11=1" "10=0,11=1" "10=1,11=0" "10=0,11=0"; do echo "knobs $kn"; MDS_KNOBS="$kn" python tools/predict_profile.py 400 1 | head -1; done
echo chunk8; python tools/predict_profile.py 800 8 | head -1
python tools/predict_timeline.py 1 > gpurun_out/predict_timeline.txt 2>&1; tail -1 gpurun_out/predict_timeline.txt
