"""14 steps of config 4's training step (4 untimed), for tools/probe/train_stats.sh."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, '/root/repo')
from vtaco_amd.bench_util import build_train_case
dev = torch.device('cuda:0')
model, trainer, batch, vf = build_train_case(dev, 0, scenes=8, grad_sync=False)
np.random.seed(0)
for _ in range(4):
    trainer.train_step(batch, vf)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    trainer.train_step(batch, vf)
torch.cuda.synchronize()
print(f"train_step: {1e2 * (time.perf_counter() - t0):.2f} ms per step (8 scenes, shipped VTacO model, pretrained t2d)")
