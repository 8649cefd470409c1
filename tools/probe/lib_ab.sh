# A/B of two builds of the library on the encoder timeline: the product build against variants/lib_head.so
# (git stash; bash tools/build_variant.sh head ""; git stash pop), twice each, alternating -- run-to-run spread on one box is ~5 %.
cd /tmp && export TMPDIR=/tmp
for K in new head new head; do
  if [ $K = new ]; then unset VTACO_HIP_LIB; else export VTACO_HIP_LIB=/root/repo/variants/lib_head.so; fi
  rm -rf /tmp/et
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/et -o e -- python3 /root/repo/tools/enc_timeline.py > /dev/null 2>&1
  echo "== $K"
  python3 /root/repo/tools/enc_timeline.py /tmp/et/e_kernel_trace.csv | grep -E "conv3d_gcr_hw|kernel time" | cut -c1-100
done
