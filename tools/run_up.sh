python -m pytest tests/test_fusion_gpu.py -q -x 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fp -o f -- python3 /root/repo/tools/bench_extra.py fusion > /tmp/fus.jsonl 2>/dev/null
grep -v amdgpu /tmp/fus.jsonl | cut -c1-200
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('/tmp/fp/f_kernel_stats.csv')))
for r in rows[:7]: print(r['Name'][:80], r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
