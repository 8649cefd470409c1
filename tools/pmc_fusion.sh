# rocprofv3 evidence for the TransformerFusion (BASELINE config 3) kernels: per-kernel time (--kernel-trace --stats) and bounded
# --pmc passes (no trace domain besides --kernel-trace) over `tools/bench_extra.py fusion` = AttentionDecoder.forward_img over the
# 128^3 lattice in 1024 chunks of N = 2048 (256 chunks per call).  Usage (GPU box): TAG=r03 bash tools/pmc_fusion.sh
cd /tmp && export TMPDIR=/tmp
R=/root/repo; TAG=${TAG:-r04}; O=$R/gpurun_out/prof_fusion_$TAG; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o fusion -- python3 $R/tools/bench_extra.py fusion > $O/bench_extra_under_profiler.jsonl 2> $O/stats.err; echo "stats rc=$?"
find $O/stats -name '*kernel_trace*' -delete 2>/dev/null
echo "kernel,counter,launches,mean_per_launch" > $O/pmc_summary.csv
pmc(){ tag=$1; shift; d=$O/pmc_$tag; timeout 600 rocprofv3 --pmc "$@" --kernel-trace -d $d -o p -- python3 $R/tools/bench_extra.py fusion > /dev/null 2>&1; echo "pmc $tag rc=$?"
  for k in expsum attend proj scalev inorm; do python3 $R/tools/pmc_summary.py fusion_$k=$d --kernel "fusion_${k}" | tail -n +2 >> $O/pmc_summary.csv; done
  rm -rf $d; }
pmc sq GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES
pmc sq2 SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
timeout 300 python3 $R/tools/bench_extra.py fusion > $O/bench_extra.jsonl 2> $O/bench_extra.err
cat $O/pmc_summary.csv; cat $O/bench_extra.jsonl
find $O/stats -name '*stats*' | head; du -sh $O
