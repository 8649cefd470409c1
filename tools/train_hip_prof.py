"""The part of BASELINE config 4's training step that runs on this repository's kernels (bench.py `train_step.ms_hip_part`:
Trainer(with_img=False, encode_t2d=False).train_step on the shipped VTacO model, 8 scenes x 2048 points): N steps after three
untimed ones.  Under `rocprofv3 --kernel-trace --stats` this gives the steady-state kernel table of that part
(profiles/rNN_train_kernel_stats.csv); alone it prints ms per step.   python tools/train_hip_prof.py [steps]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vtaco_amd.bench_util import build_train_case  # noqa: E402
from vtaco_amd.conv_onet.training import Trainer  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
model, trainer, batch, vf = build_train_case(dev, 0, scenes=8, pretrained_t2d=True, grad_sync=False)
vis = Trainer(model, trainer.optimizer, device=dev, input_type="pointcloud", threshold=0.5, num_sample=2048, with_img=False, encode_t2d=False)
np.random.seed(0)
for _ in range(3):
    vis.train_step(batch, vf)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    vis.train_step(batch, vf)
torch.cuda.synchronize()
print(f"{1e3 * (time.perf_counter() - t0) / steps:.2f} ms per step over {steps} steps (visual + hand branches, forward + backward + Adam)")
