"""Marching cubes alone at 128^3 and 256^3 on a decoded scene: p50 of vt_mc_count + read-back + vt_mc_emit (ops.marching_cubes),
and per-kernel times when run under rocprofv3 --kernel-trace --stats."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vtaco_amd import ops  # noqa: E402
from vtaco_amd.bench_util import build_scene  # noqa: E402

dev = torch.device("cuda:0")
sc = build_scene(0, dev)
for nx in (128, 256):
    vol = sc["model"].decoder.decode_lattice(sc["grid"], nx, precision="f16x3").view(nx, nx, nx)
    ts = []
    for i in range(60):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        v, f = ops.marching_cubes(vol, 0.2, rescale=(nx // 2, 1.1 / nx))[:2]
        torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
    ts = sorted(ts[10:])
    print(f"{nx}^3: p50 {ts[len(ts) // 2]:.3f} ms  min {ts[0]:.3f} ms  verts {v.shape[0]} faces {f.shape[0]}")
