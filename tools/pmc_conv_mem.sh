cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/pmc_conv_mem; mkdir -p $O
echo "kernel,counter,launches,mean_per_launch" > $O/summary.csv
pmc(){ tag=$1; shift; d=$O/$tag; timeout 300 rocprofv3 --pmc "$@" --kernel-trace -d $d -o p -- python3 $R/tools/enc_timeline.py > /dev/null 2>&1; echo "pmc $tag rc=$?"
  python3 $R/tools/pmc_summary.py conv_h8=$d --kernel "conv3d_gcr_hw_kernelILi8E,conv3d_gcr_hw_kernel<8>" | tail -n +2 >> $O/summary.csv
  python3 $R/tools/pmc_summary.py conv_h4=$d --kernel "conv3d_gcr_hw_kernelILi4E,conv3d_gcr_hw_kernel<4>" | tail -n +2 >> $O/summary.csv
  rm -rf $d; }
pmc f FETCH_SIZE
pmc w WRITE_SIZE
pmc t TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
cat $O/summary.csv
