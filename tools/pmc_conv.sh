# Counters of the UNet3D conv kernels (three --pmc passes over a short bench run; per-launch means per kernel family):
# conv_h8 = conv3d_gcr_hw_kernel<8> (64^3 level), conv_h4 = <4> (32^3 level), conv_s2 = conv3d_gcr_s_kernel<2> (16^3 level),
# conv_up8 / conv_up4 = conv3d_gcr_up_kernel<8> / <4> (the decoder-entry layers of the 64^3 / 32^3 levels in per-parity form).
# Usage (GPU box): bash tools/pmc_conv.sh  ->  gpurun_out/pmc_conv/summary.csv
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/pmc_conv; mkdir -p $O
echo "kernel,counter,launches,mean_per_launch" > $O/summary.csv
pmc(){ tag=$1; shift; d=$O/$tag; timeout 300 rocprofv3 --pmc "$@" --kernel-trace -d $d -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train > /dev/null 2>&1; echo "pmc $tag rc=$?"
  python3 $R/tools/pmc_summary.py conv_h8=$d --kernel "conv3d_gcr_hw_kernelILi8E,conv3d_gcr_hw_kernel<8>" | tail -n +2 >> $O/summary.csv
  python3 $R/tools/pmc_summary.py conv_h4=$d --kernel "conv3d_gcr_hw_kernelILi4E,conv3d_gcr_hw_kernel<4>" | tail -n +2 >> $O/summary.csv
  python3 $R/tools/pmc_summary.py conv_s2=$d --kernel "conv3d_gcr_s_kernelILi2E,conv3d_gcr_s_kernel<2>" | tail -n +2 >> $O/summary.csv
  python3 $R/tools/pmc_summary.py conv_up8=$d --kernel "conv3d_gcr_up_kernelILi8E,conv3d_gcr_up_kernel<8>" | tail -n +2 >> $O/summary.csv
  python3 $R/tools/pmc_summary.py conv_up4=$d --kernel "conv3d_gcr_up_kernelILi4E,conv3d_gcr_up_kernel<4>" | tail -n +2 >> $O/summary.csv
  rm -rf $d; }
pmc a GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES
pmc b SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pmc c SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU
cat $O/summary.csv
