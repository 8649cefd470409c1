# A/B of the pooled levels' statistics: VTACO_POOL_FUSED_VOX = 0 (max-pool + statistics launches), 2, 4, 8 voxels per block of the fused kernel
export TMPDIR=/tmp; cd /root/repo
for v in 0 2 4 8; do
  export VTACO_POOL_FUSED_VOX=$v
  (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /root/repo/gpurun_out/pool$v -o e -- python3 /root/repo/tools/enc_timeline.py > /dev/null 2>&1)
  python3 tools/enc_timeline.py gpurun_out/pool$v/e_kernel_trace.csv > gpurun_out/pool_ab_$v.txt
  echo "== VTACO_POOL_FUSED_VOX=$v"; grep -E "maxpool|channel_stats|kernel time" gpurun_out/pool_ab_$v.txt | cut -c1-46,60-80
  rm -rf gpurun_out/pool$v
done
python -m pytest tests/test_unet3d_gpu.py tests/test_encoder_gpu.py -x -q -m gpu 2>&1 | tail -2
