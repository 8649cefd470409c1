cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/prof_r01e; mkdir -p $O
echo "kernel,counter,launches,mean_per_launch" > $O/pmc_summary.csv
for P in bf16x3 f32; do
  if [ $P = bf16x3 ]; then KPAT='staged_kernelILb1E,staged_kernel<true>'; else KPAT='staged_kernelILb0E,staged_kernel<false>'; fi
  pmc(){ tag=$1; shift; d=$O/pmc_${P}_$tag; timeout 200 rocprofv3 --pmc "$@" --kernel-trace -d $d -o p -- python3 $R/bench.py --steps 10 --warmup 2 --decode-only --no-cpu-baseline --precision $P > /dev/null 2>&1; echo "pmc $P $tag rc=$?"; python3 $R/tools/pmc_summary.py decode_$P=$d --kernel "$KPAT" | tail -n +2 >> $O/pmc_summary.csv; rm -rf $d; }
  pmc fetch FETCH_SIZE
  pmc write WRITE_SIZE
  pmc sq GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES
  pmc sq2 SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS
done
cat $O/pmc_summary.csv
