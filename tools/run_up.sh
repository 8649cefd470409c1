cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_unet3d_gpu.py -x -q -k "per_parity or vs_oracle or fold or accumulator" 2>&1 | tail -15
VTACO_CONV_UP=1 bash tools/probe/enc_tl.sh up1 | grep -E "up_kernel|ksum|kernel time"
VTACO_CONV_UP_KSPLIT=0 bash tools/probe/enc_tl.sh up1 | grep -E "up_kernel|ksum|s_kernel<8|kernel time"
VTACO_CONV_UP=1 bash tools/probe/enc_tl.sh up1 | grep -E "up_kernel|ksum|kernel time"
