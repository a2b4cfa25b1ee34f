export TMPDIR=/tmp; rm -rf /tmp/pkt; mkdir -p /tmp/pkt
python tools/predict_profile.py 400 1
(cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/pkt -o kt --output-format csv -- python /root/repo/tools/predict_profile.py 400 1 > /dev/null 2>&1)
f=$(find /tmp/pkt -name '*kernel_stats.csv' | head -1); cp $f gpurun_out/predict_fbf_kernel_stats.csv
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/predict_fbf_kernel_stats.csv')))
n=400*2+64
tot=0
for r in rows[:40]:
    t=float(r['TotalDurationNs'])/n/1e3; tot+=t
    print(f"{r['Name'][:90]:90s} {int(r['Calls'])/n:6.2f}/frame {t:8.1f} us/frame avg {float(r['AverageNs'])/1e3:7.1f}")
print('sum all', sum(float(r['TotalDurationNs']) for r in rows)/n/1e3, 'launches/frame', sum(int(r['Calls']) for r in rows)/n)
PY
