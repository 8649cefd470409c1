"""Framework glue of the HIP part of config 4's training step: aten ops by count and input shape (torch.profiler, CPU side), to find
the copies / transposes / reductions that a kernel-side change would remove."""
import sys
import numpy as np
import torch
from torch.profiler import profile, ProfilerActivity
sys.path.insert(0, '/root/repo')
from vtaco_amd.bench_util import build_train_case
from vtaco_amd.conv_onet.training import Trainer
dev = torch.device('cuda:0')
model, trainer, batch, vf = build_train_case(dev, 0, scenes=8, pretrained_t2d=True, grad_sync=False)
vis = Trainer(model, trainer.optimizer, device=dev, input_type="pointcloud", threshold=0.5, num_sample=2048, with_img=False, encode_t2d=False)
np.random.seed(0)
for _ in range(3):
    vis.train_step(batch, vf)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    vis.train_step(batch, vf)
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith("aten::") and e.device_time_total > 0]
rows.sort(key=lambda e: -e.device_time_total)
for e in rows[:60]:
    print(f"{e.device_time_total / 1e3:7.3f} ms x{e.count:4d}  {e.key:28s} {str(e.input_shapes)[:110]}")
