"""Host side of the HIP part of config 4's training step: cProfile of steady steps (where the Python / ctypes / allocator time goes
when the step is launch-bound)."""
import cProfile, pstats, sys, time
import numpy as np
import torch
sys.path.insert(0, '/root/repo')
from vtaco_amd.bench_util import build_train_case
from vtaco_amd.conv_onet.training import Trainer
dev = torch.device('cuda:0')
model, trainer, batch, vf = build_train_case(dev, 0, scenes=8, pretrained_t2d=True, grad_sync=False)
vis = Trainer(model, trainer.optimizer, device=dev, input_type="pointcloud", threshold=0.5, num_sample=2048, with_img=False, encode_t2d=False)
np.random.seed(0)
for _ in range(4):
    vis.train_step(batch, vf)
torch.cuda.synchronize()
# host time alone: steps enqueued without waiting for the GPU
t0 = time.perf_counter()
for _ in range(5):
    vis.train_step(batch, vf)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"5 steps: enqueue {1e3 * (t1 - t0) / 5:.2f} ms per step, drained after another {1e3 * (t2 - t1):.2f} ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    vis.train_step(batch, vf)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumulative").print_stats("vtaco_amd|training", 30)
