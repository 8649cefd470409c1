"""Which framework glue ops (copies, adds, fills) run in the HIP part of config 4's training step, by call site.
GPU box: python3 tools/probe/train_glue_ops.py"""
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
for _ in range(4):
    vis.train_step(batch, vf)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    vis.train_step(batch, vf)
    torch.cuda.synchronize()
want = ("aten::copy_", "aten::add_", "aten::add", "aten::fill_", "aten::zero_", "aten::mul", "aten::cat", "aten::sum", "aten::div", "aten::sub",
        "aten::clone", "aten::contiguous", "aten::index", "aten::mean", "aten::neg", "aten::where", "aten::_to_copy")
rows = []
for e in prof.events():
    if e.device_type != torch.autograd.DeviceType.CPU or e.name not in want:
        continue
    dt = sum(k.duration for k in e.kernels) if e.kernels else 0.0
    if not e.kernels:
        continue
    site = "?"
    for fr in (e.stack or []):
        if "/root/repo/" in fr and "probe" not in fr:
            site = fr.replace("/root/repo/", "")
            break
    rows.append((e.name, str(e.input_shapes)[:70], site[:110], dt))
agg = {}
for name, shp, site, dt in rows:
    k = (name, shp, site)
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1; a[1] += dt
tot = sum(v[1] for v in agg.values())
print(f"glue ops with kernels: {sum(v[0] for v in agg.values())} calls, {tot / 1e3:.3f} ms of device time")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
    print(f"{v[1]:8.1f} us x{v[0]:3d}  {k[0]:16s} {k[1]:70s} {k[2]}")
