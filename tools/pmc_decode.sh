cd /tmp && export TMPDIR=/tmp
R=/root/repo
run(){ tag=$1; lib=$2; shift 2; VTACO_HIP_LIB=$lib rocprofv3 --pmc "$@" --kernel-trace -d $R/gpurun_out/pmc_$tag -o p -- python3 $R/bench.py --steps 20 --warmup 5 --decode-only --no-cpu-baseline --precision bf16x3 > /dev/null 2>&1; }
for v in nomlp base; do
  L=$R/vtaco_amd/variants/lib_$v.so
  run ${v}_a $L TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr
  run ${v}_b $L TA_TOTAL_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
  run ${v}_c $L GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES
  run ${v}_d $L SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_INSTS_LDS
done
ls $R/gpurun_out/ | head -30
