# Counters of the split-half weight-gradient kernel (three --pmc passes over tools/bench_wgrad.py's first shape only).
# Usage (GPU box): bash tools/pmc_wgrad.sh  ->  gpurun_out/pmc_wgrad/summary.csv
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/pmc_wgrad; mkdir -p $O
echo "kernel,counter,launches,mean_per_launch" > $O/summary.csv
pmc(){ tag=$1; shift; d=$O/$tag; timeout 300 rocprofv3 --pmc "$@" --kernel-trace -d $d -o p -- python3 $R/tools/bench_wgrad.py --first > /dev/null 2>&1; echo "pmc $tag rc=$?"
  python3 $R/tools/pmc_summary.py wgrad_h=$d --kernel "conv3d_wgrad_h_kernel" | tail -n +2 >> $O/summary.csv
  rm -rf $d; }
pmc a GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES
pmc b SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pmc c SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU
pmc d SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM
cat $O/summary.csv
