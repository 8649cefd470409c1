# Collect the round's rocprofv3 evidence on the GPU box into gpurun_out/prof_$TAG (every run bounded by its own timeout;
# --pmc passes carry no trace domain besides --kernel-trace; counter databases are summarised and deleted on the box:
# gpurun copies back at most 64 MiB).  Usage: TAG=r01e bash tools/collect_profiles.sh
cd /tmp && export TMPDIR=/tmp
R=/root/repo
FAILED=0
# every step reports its exit status; a failed step marks the whole collection failed (exit 1 at the end) and its output is not
# copied anywhere: profiles/ is only written through tools/keep_evidence.py, which refuses empty files and tracebacks
step(){ name=$1; rc=$2; if [ "$rc" != 0 ]; then echo "FAILED: $name rc=$rc"; FAILED=1; else echo "$name rc=0"; fi; }
TAG=${TAG:-r06}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_under_profiler.json 2> $O/stats.err; step stats $?
find $O/stats -name '*kernel_trace*' -delete 2>/dev/null
# the headline kernel alone (128^3 launches only: the full bench above also decodes 256^3 slabs with the same kernel, which skews its average)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_decode -o decode -- python3 $R/bench.py --steps 200 --warmup 20 --decode-only --no-cpu-baseline > $O/bench_decode_under_profiler.json 2> $O/stats_decode.err; step "decode stats" $?
find $O/stats_decode -name '*kernel_trace*' -delete 2>/dev/null
echo "# decode_sources=$(python3 $R/tools/src_hash.py)" > $O/pmc_summary.csv
echo "kernel,counter,launches,mean_per_launch" >> $O/pmc_summary.csv
for P in f16x3 f16f8 bf16x3 f32; do
  # the bench also launches the other precision's kernel (its exact-f32 side measurement): keep one kernel per key
  case $P in f16f8) KPAT='staged3_kernelILi2E,staged3_kernel<2>';; f16x3) KPAT='staged3_kernelILi1E,staged3_kernel<1>';;
             bf16x3) KPAT='staged2_kernelILi1E,staged2_kernel<1>';; *) KPAT='staged2_kernelILi0E,staged2_kernel<0>';; esac
  pmc(){ tag=$1; shift; d=$O/pmc_${P}_$tag; timeout 200 rocprofv3 --pmc "$@" --kernel-trace -d $d -o p -- python3 $R/bench.py --steps 10 --warmup 2 --decode-only --no-cpu-baseline --precision $P > /dev/null 2>&1; step "pmc $P $tag" $?; python3 $R/tools/pmc_summary.py decode_$P=$d --kernel "$KPAT" | tail -n +2 >> $O/pmc_summary.csv; rm -rf $d; }
  pmc fetch FETCH_SIZE
  pmc write WRITE_SIZE
  pmc sq GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES
  pmc sq2 SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES
done
timeout 300 python3 $R/bench.py > $O/bench.json 2> $O/bench.err; step bench $?
timeout 600 python3 $R/tools/bench_extra.py > $O/bench_extra.jsonl 2> $O/bench_extra.err; step extra $?
du -sh $O
exit $FAILED
