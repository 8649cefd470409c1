# config 4's training step with the tactile feature encoder on a side stream (default, single process) against one stream.  GPU box.
for ov in 1 0 1 0; do echo "VTACO_TRAIN_OVERLAP=$ov: $(VTACO_TRAIN_OVERLAP=$ov python3 tools/probe/train_hip_step.py full 2>&1 | grep 'ms per step')"; done
