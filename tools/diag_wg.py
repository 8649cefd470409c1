"""Per-workgroup lifetimes of one lattice decode launch (the stamps every launch leaves: ops.decode_last_clock): where the time
between "workgroup 0 is done" and "the kernel is done" goes.  python tools/diag_wg.py [precision] [nx]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vtaco_amd import ops
from vtaco_amd.bench_util import build_scene
prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
nx = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = torch.device("cuda:0")
sc = build_scene(0, dev)
dec, grid = sc["model"].decoder, sc["grid"]
out = torch.empty((1, nx ** 3), dtype=torch.float32, device=dev)
for _ in range(2000):
    dec.decode_lattice(grid, nx, out=out, precision=prec)
torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for _ in range(200):
    dec.decode_lattice(grid, nx, out=out, precision=prec)
ev1.record(); torch.cuda.synchronize()
c = ops.decode_last_clock(workgroups=True)
t = np.array(c.pop("wg_ticks"), dtype=np.float64) * 1e3 / c["ref_khz"]
print(f"{prec} {nx}^3: {ev0.elapsed_time(ev1) / 200 * 1e3:.1f} us per launch back to back; last launch:", {k: (round(v, 2) if isinstance(v, float) else v) for k, v in c.items()})
print("per XCD (workgroup b runs on XCD b & 7): start min/max, end min/max, duration median [us]")
for x in range(8):
    r = t[x::8]
    print(f"  xcd {x}: start {r[:,0].min():7.2f} {r[:,0].max():7.2f}   end {r[:,1].min():7.2f} {r[:,1].max():7.2f}   dur {np.median(r[:,1]-r[:,0]):7.2f}  max {np.max(r[:,1]-r[:,0]):7.2f}")
order = np.argsort(t[:, 1])
print("the ten workgroups that end last:", [(int(i), round(t[i, 0], 1), round(t[i, 1], 1)) for i in order[-10:]])
print("the ten that end first:", [(int(i), round(t[i, 0], 1), round(t[i, 1], 1)) for i in order[:10]])
