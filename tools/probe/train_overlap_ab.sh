# config 4's training step with the hand branch and the tactile feature encoder on side streams (default, single process) against one
# stream, the part on this repository's kernels and the whole step.  GPU box.
for ov in 1 0 1 0; do echo "VTACO_TRAIN_OVERLAP=$ov: hip part $(VTACO_TRAIN_OVERLAP=$ov python3 tools/probe/train_hip_step.py 2>&1 | grep 'ms per step'), whole $(VTACO_TRAIN_OVERLAP=$ov python3 tools/probe/train_hip_step.py full 2>&1 | grep 'ms per step')"; done
