# encoder timeline with and without the first layer's empty blocks skipped (VTACO_UNET_SKIP), same box
cd /root/repo
for S in 0 1 0 1; do
  echo "== VTACO_UNET_SKIP=$S"
  VTACO_UNET_SKIP=$S bash tools/probe/enc_tl.sh skip$S | grep -E "hw_kernel|tile_flags|kernel time"
done
