# kernel statistics of config 4's training step (tools/train_prof.py without its cProfile part would do; this is the short form)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ts -o t -- python3 /root/repo/tools/probe/train_loop.py > /tmp/ts.out 2>&1
tail -2 /tmp/ts.out
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/ts/**/t_kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"total kernel time {tot / 1e6:.1f} ms")
for r in rows[:22]:
    print(f"{float(r['TotalDurationNs']) / 1e6:9.2f} ms {int(r['Calls']):6d} calls  {r['Name'][:110]}")
PY
cp $(find /tmp/ts -name 't_kernel_stats.csv') /root/repo/gpurun_out/train_loop_kernel_stats.csv
