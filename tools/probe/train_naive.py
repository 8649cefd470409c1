"""Which framework ops of config 4's training step end up on MIOpen's naive convolution kernels?  One step under torch.profiler with
shapes; prints the aten convolution calls sorted by device time with their input shapes."""
import sys
import numpy as np
import torch
from torch.profiler import profile, ProfilerActivity
sys.path.insert(0, '/root/repo')
from vtaco_amd.bench_util import build_train_case
dev = torch.device('cuda:0')
model, trainer, batch, vf = build_train_case(dev, 0, scenes=8, grad_sync=False)
np.random.seed(0)
for _ in range(3):
    trainer.train_step(batch, vf)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    trainer.train_step(batch, vf)
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if "conv" in e.key.lower()]
rows.sort(key=lambda e: -e.device_time_total)
print("-- kernels with 'naive' in the name:")
for e in prof.key_averages():
    if "naive" in e.key:
        print(f"{e.device_time_total / 1e3:9.3f} ms  x{e.count:3d}  {e.key[:90]}")
print("-- total device time of the step's kernels:", sum(e.device_time_total for e in prof.key_averages() if e.device_type == torch.autograd.DeviceType.CUDA) / 1e3, "ms")
print("-- convolution ops by device time:")
rows = [e for e in rows if e.key.startswith("aten::miopen") or e.key.startswith("aten::convolution_backward") or "naive" in e.key]
for e in rows[:24]:
    print(f"{e.device_time_total / 1e3:9.3f} ms  x{e.count:3d}  {e.key:45s} {str(e.input_shapes)[:150]}")
