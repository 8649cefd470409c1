"""One steady-state step of the part of config 4's training step that runs on this repository's kernels (tools/train_hip_prof.py's
trainer) under torch.profiler: kernels by device time, and the wall time of the step.  ``full``: the whole config-4 step (with_img, encode_t2d)."""
import sys, time
import numpy as np
import torch
from torch.profiler import profile, ProfilerActivity
sys.path.insert(0, '/root/repo')
from vtaco_amd.bench_util import build_train_case
from vtaco_amd.conv_onet.training import Trainer
dev = torch.device('cuda:0')
model, trainer, batch, vf = build_train_case(dev, 0, scenes=8, pretrained_t2d=True, grad_sync=False)
vis = trainer if len(sys.argv) > 1 and sys.argv[1] == "full" else Trainer(model, trainer.optimizer, device=dev, input_type="pointcloud", threshold=0.5, num_sample=2048, with_img=False, encode_t2d=False)
np.random.seed(0)
for _ in range(4):
    vis.train_step(batch, vf)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    vis.train_step(batch, vf)
torch.cuda.synchronize()
print(f"{1e2 * (time.perf_counter() - t0):.2f} ms per step")
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    vis.train_step(batch, vf)
    torch.cuda.synchronize()
# (kineto mirrors user annotations such as "Optimizer.step#Adam.step" onto the device timeline as ranges over their kernels: not launches)
ev = [e for e in prof.key_averages() if e.device_type == torch.autograd.DeviceType.CUDA and not getattr(e, "is_user_annotation", False)
      and not e.key.startswith("Optimizer.")]
ev.sort(key=lambda e: -e.device_time_total)
print(f"device time of the step's kernels: {sum(e.device_time_total for e in ev) / 1e3:.2f} ms over {sum(e.count for e in ev)} launches")
for e in ev[:45]:
    print(f"{e.device_time_total / 1e3:8.3f} ms x{e.count:4d}  {e.key[:120]}")
