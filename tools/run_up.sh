cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_unet3d_gpu.py -x -q -k "per_parity" 2>&1 | tail -15
bash tools/probe/up_stamps.sh
VTACO_CONV_UP=1 bash tools/probe/enc_tl.sh up1 | grep -E "up_kernel|kernel time"
VTACO_CONV_UP=1 bash tools/probe/enc_tl.sh up1 | grep -E "up_kernel|kernel time"
