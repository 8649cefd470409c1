# rocprofv3 evidence for the general-shape decoder (decode_wide.hip): per-kernel time (--kernel-trace --stats) and bounded --pmc passes
# (no trace domain besides --kernel-trace) over `tools/bench_extra.py wide` = LocalDecoder 256 / 128 / 5 and 64 / 32 / 5 over the 128^3
# lattice, exact f32 and split f16.  Usage (GPU box): TAG=r06 bash tools/pmc_wide.sh
cd /tmp && export TMPDIR=/tmp
R=/root/repo; TAG=${TAG:-r06}; O=$R/gpurun_out/prof_wide_$TAG; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o wide -- python3 $R/tools/bench_extra.py wide > $O/bench_extra_under_profiler.jsonl 2> $O/stats.err; echo "stats rc=$?"
find $O/stats -name '*kernel_trace*' -delete 2>/dev/null
echo "# wide_sources=$(python3 $R/tools/src_hash.py wide)" > $O/pmc_summary.csv
echo "kernel,counter,launches,mean_per_launch" >> $O/pmc_summary.csv
# keys (patterns match the MANGLED names the counter database holds): the exact-f32 kernel at 256 / 128 (8 waves) and 64 / 32 (4 waves), the split-f16 streaming kernel (256 / 128), the 64 / 32
# register-resident pipeline and its sampling pre-pass
pmc(){ tag=$1; shift; d=$O/pmc_$tag; timeout 600 rocprofv3 --pmc "$@" --kernel-trace -d $d -o p -- python3 $R/tools/bench_extra.py wide > /dev/null 2>&1; echo "pmc $tag rc=$?"
  for kv in "wide_f32_256=decode_wide_kernelILi8E" "wide_f32_64=decode_wide_kernelILi4E" "wide_f16x3_256=decode_wide_h_kernel" "wide_f16x3_64=decode_wide_p_kernel" "wide_sample=wide_sample_kernel"; do
    python3 $R/tools/pmc_summary.py ${kv%%=*}=$d --kernel "${kv#*=}" | tail -n +2 >> $O/pmc_summary.csv; done
  rm -rf $d; }
pmc sq GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES
pmc sq2 SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc l2 TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum
timeout 300 python3 $R/tools/bench_extra.py wide > $O/bench_extra.jsonl 2> $O/bench_extra.err
cat $O/pmc_summary.csv; cat $O/bench_extra.jsonl
find $O/stats -name '*stats*' | head; du -sh $O
