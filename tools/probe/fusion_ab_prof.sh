# kernel-level A/B of the fusion passes under rocprofv3 (per-kernel averages of tools/bench_extra.py fusion): $1 = env var, $2.. = values
cd /tmp && export TMPDIR=/tmp
R=/root/repo; VAR=$1; shift
for v in "$@"; do
  O=$R/gpurun_out/fab_$v; rm -rf $O; mkdir -p $O
  export $VAR=$v
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o f -- python3 $R/tools/bench_extra.py fusion > $O/out.txt 2>&1
  echo "== $VAR=$v: $(cut -c1-120 $O/out.txt | tail -1)"
  python3 - $O <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
tot = 0
for r in list(csv.DictReader(open(f)))[:9]:
    n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:48]
    print(f"  {n:50s} {r['Calls']:>4s} x {float(r['AverageNs'])/1e3:8.1f} us")
PY
  find $O -name '*kernel_trace*' -delete
done
