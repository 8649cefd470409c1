# A/B of the K-split plans of the thin UNet3D levels: one encoder timeline per forced (tile depth, slices) pair.
# The knob is read before the profiler's exec hop matters (plain environment variable of the profiled python).
cd /tmp && export TMPDIR=/tmp
for K in default 0 2,2 2,4 2,8 8,2 8,4 8,8; do
  if [ $K = default ]; then unset VTACO_CONV_KSPLIT; else export VTACO_CONV_KSPLIT=$K; fi
  rm -rf /tmp/et
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/et -o e -- python3 /root/repo/tools/enc_timeline.py > /dev/null 2>&1
  echo "== VTACO_CONV_KSPLIT=$K"
  python3 /root/repo/tools/enc_timeline.py /tmp/et/e_kernel_trace.csv | sed -n '26,52p;$p' | grep -v "gn_finalize\|maxpool\|channel_stats" | cut -c1-86
done
