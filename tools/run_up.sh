cd $GRAFT_REPO_ROOT
for V in 1 0 1 0; do echo "== VTACO_CONV_UP_TRAIN=$V"; VTACO_CONV_UP_TRAIN=$V timeout 600 python tools/train_hip_prof.py 20 2>&1 | tail -1; done
