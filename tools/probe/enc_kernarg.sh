# encoder timeline with kernel arguments in device memory or not (HIP_FORCE_DEV_KERNARG), same box
cd /root/repo
for K in 0 1 0 1; do
  echo "== HIP_FORCE_DEV_KERNARG=$K"
  HIP_FORCE_DEV_KERNARG=$K bash tools/probe/enc_tl.sh ka$K | grep -E "kernel time"
done
echo "== unset"; bash tools/probe/enc_tl.sh kau | grep -E "kernel time"
