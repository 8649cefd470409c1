"""Device time of the voxel sort's launch in its three forms (plain, + clearing the mean grid, + block flags) on the bench cloud.
GPU box: python3 tools/probe/voxel_build_time.py"""
import os, sys
import torch
from torch.profiler import profile, ProfilerActivity
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vtaco_amd import ops
from vtaco_amd.bench_util import sphere_cloud
dev = torch.device("cuda:0")
for B, T in ((1, 3000), (1, 2048), (8, 3000)):
    p = torch.cat([sphere_cloud(i, T) for i in range(B)], 0).to(dev)
    grid = torch.empty((B, 64, 64, 64, 32), device=dev)
    forms = {"plain": lambda: ops.VoxelIndex(p, 64, 0.1),
             "+ clear 33 MB/scene": lambda: ops.VoxelIndex(p, 64, 0.1, clear=grid),
             "+ clear + block flags": lambda: ops.VoxelIndex(p, 64, 0.1, clear=grid, want_tile_flags=True),
             "plane 32^2": lambda: ops.PlaneIndex(p, 32, 0.1, "xz")}
    for name, fn in forms.items():
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
        ev = [e for e in prof.key_averages() if e.device_type == torch.autograd.DeviceType.CUDA and "voxel_build" in e.key]
        print(f"B={B} T={T} {name:24s} {sum(e.device_time_total for e in ev) / sum(e.count for e in ev):7.1f} us per launch")
