python -m pytest tests/test_unet3d_gpu.py tests/test_train_gpu.py tests/test_encoder_gpu.py -q -x 2>&1 | tail -3
for v in 1 0 1 0; do echo "BWD_OVERLAP=$v"; VTACO_UNET_BWD_OVERLAP=$v python3 tools/probe/train_hip_step.py 2>&1 | grep -E "ms per step|device time"; done
