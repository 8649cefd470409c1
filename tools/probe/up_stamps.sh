# Where the per-parity decoder-entry conv's time goes (stamped build: tools/build_variant.sh hb "-DVT_DIAG_HB").
cd /root/repo
O=gpurun_out/up_stamps.txt
: > $O
for SHAPE in "64 32 64 32" "32 64 128 64"; do
  echo "######## per-parity kernel, R C1 C2 Cout = $SHAPE" >> $O
  VTACO_HIP_LIB=variants/lib_hb.so timeout 200 python3 tools/diag_conv_up.py $SHAPE 2>&1 | grep -v amdgpu.ids >> $O
done
cat $O
