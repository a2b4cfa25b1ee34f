python -m pytest tests/test_predictor.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|assert|^E " | head -20
python tools/predict_profile.py 400 1
python tools/predict_profile.py 400 1 bf16 | head -1
echo chunk8; python tools/predict_profile.py 800 8 | head -1
python tools/predict_timeline.py 1 > gpurun_out/predict_timeline.txt 2>&1; tail -1 gpurun_out/predict_timeline.txt
