for v in 1 0 1 0; do echo "MASK_FUSE=$v"; VTACO_UNET_MASK_FUSE=$v python3 tools/probe/train_hip_step.py 2>&1 | grep -E "ms per step|device time"; done
