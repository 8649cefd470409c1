set -x
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_unet3d_gpu.py -x -q -k "per_parity or split_f16_conv_layers" 2>&1 | tail -15
