#!/bin/bash
# A/B of the split-f16 lattice decode kernels on the bench's decode timing: the compiler-scheduled double-brick kernel
# (VTACO_DECODE_ST3=0), the slot-pipelined one with the 5-instruction split (default) and with the 4-instruction one; extra
# arguments are variant builds (variants/lib_NAME.so) run with the default settings.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/st3_ab; mkdir -p $O
run() { env "$@" python3 bench.py --steps 200 --warmup 20 --decode-only --no-cpu-baseline $BENCH_ARGS 2>>$O/err.txt | python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f ms  %.3e pts/s' % (j['roofline']['kernel_ms'], j['value']))"; }
for r in 1 2 3; do
  echo "round $r"
  echo -n "  staged2           : "; BENCH_ARGS="--precision f16x3" run VTACO_DECODE_ST3=0
  echo -n "  staged3 split5    : "; BENCH_ARGS="--precision f16x3" run VTACO_DECODE_ST3=1
  echo -n "  staged3 split4mix : "; BENCH_ARGS="--precision f16x3" run VTACO_DECODE_ST3=1 VTACO_DECODE_ST3_SPLIT=0
  echo -n "  f16f8             : "; BENCH_ARGS="--precision f16f8" run VTACO_DECODE_ST3=1
  for v in "$@"; do printf "  %-18s: " "$v"; run VTACO_HIP_LIB=variants/lib_$v.so; done
done | tee $O/ab.txt
