# Where a 64^3 split-f16 conv layer's time goes: the stamped build of both schedules (hw: tap + loader waves, hx: the staging in the tap
# waves' MFMA gaps) and the timing-only ablations of hx.  Build first: tools/build_variant.sh hb "-DVT_DIAG_HB" and
# tools/build_variant.sh ablN "-DVT_DIAG_HB -DVT_HX_ABL=N" for N in 1 2 3 8 16 32.
cd /root/repo
O=gpurun_out/conv_stamps.txt
: > $O
for SHAPE in "64 32 32" "64 96 32"; do
  echo "######## specialised waves (default), R C1 Cout = $SHAPE" >> $O
  VTACO_CONV_SPEC=1 VTACO_HIP_LIB=variants/lib_hb.so timeout 200 python3 tools/diag_conv.py $SHAPE 2>&1 | grep -v amdgpu.ids >> $O
  echo "######## staging in the tap waves' gaps (VTACO_CONV_SPEC=2), R C1 Cout = $SHAPE" >> $O
  VTACO_CONV_SPEC=2 VTACO_HIP_LIB=variants/lib_hb.so timeout 200 python3 tools/diag_conv.py $SHAPE 2>&1 | grep -v amdgpu.ids >> $O
done
for A in 1 2 3 8 16 32; do
  echo "######## VTACO_CONV_SPEC=2 ablation VT_HX_ABL=$A (1 no staging pieces, 2 no chunk barrier, 8 no commit, 16 no weight DMA, 32 no requests; results wrong, timing only), 64 32 32" >> $O
  VTACO_CONV_SPEC=2 VTACO_HIP_LIB=variants/lib_abl$A.so timeout 200 python3 tools/diag_conv.py 64 32 32 2>&1 | grep -E "counter|taps \+|chunk barrier  |epilogue|TOTAL" >> $O
done
cat $O
