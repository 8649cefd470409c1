# PMC passes over the decode bench (each bounded by its own timeout; counters only, no trace domains besides kernel-trace)
cd /tmp && export TMPDIR=/tmp
R=/root/repo
PREC=${PREC:-bf16x3}
run(){ tag=$1; shift; timeout 150 rocprofv3 --pmc "$@" --kernel-trace -d $R/gpurun_out/pmc_$tag -o p -- python3 $R/bench.py --steps 20 --warmup 5 --decode-only --no-cpu-baseline --precision $PREC > /dev/null 2>&1; echo "$tag rc=$?"; }
run ${PREC}_sq1 GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES
run ${PREC}_sq2 SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY
run ${PREC}_sq3 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU
run ${PREC}_mem FETCH_SIZE WRITE_SIZE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
