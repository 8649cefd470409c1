timeout 900 python -m pytest tests/test_decode_wide_gpu.py -q -x 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wp -o w -- python3 /root/repo/tools/bench_extra.py wide > /tmp/wide.jsonl 2>/dev/null
cut -c1-230 /tmp/wide.jsonl | grep -v amdgpu
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('/tmp/wp/w_kernel_stats.csv')))
for r in rows[:8]: print(r['Name'][:90], r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
