# encoder timeline A/B over VTACO_CONV_UP (per-parity decoder-entry convs) on one box.  Usage: bash tools/probe/enc_up_ab.sh
cd /root/repo
for S in 1 0 1 0; do
  echo "== VTACO_CONV_UP=$S"
  VTACO_CONV_UP=$S bash tools/probe/enc_tl.sh up$S | grep -E "hw_kernel|up_kernel|s_kernel|kernel time"
done
