# encoder timeline A/B over VTACO_CONV_SPEC on one box.  Usage: bash tools/probe/enc_ab.sh
cd /root/repo
for S in 1 2 1 2; do
  echo "== VTACO_CONV_SPEC=$S"
  VTACO_CONV_SPEC=$S bash tools/probe/enc_tl.sh ab$S | grep -E "hw_kernel|hx_kernel|kernel time"
done
