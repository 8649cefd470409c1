"""Diagnostic (VT_DIAG_CLOCK build only): in-kernel shader clock of the decode kernel."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vtaco_amd import ops
from vtaco_amd.bench_util import build_scene
dev = torch.device("cuda:0")
sc = build_scene(0, dev)
dec, grid = sc["model"].decoder, sc["grid"]
nx = 128
for _ in range(200):
    dec.decode_lattice(grid, nx)
torch.cuda.synchronize()
blob = dec._blob(contact=False)
out, out2 = ops.decode_fwd(grid, blob, lattice=(nx, 1.1, 0, nx ** 3), want_contact=True)
torch.cuda.synchronize()
raw = out2.reshape(-1).view(torch.int64)[:2048].cpu().view(-1, 4)
d = raw[:, :2].double()
start = (raw[:, 2] - raw[:, 2].min()).double() / 100.0
dur = d[:, 1] / 100.0
import numpy as np
print("start us percentiles", np.percentile(start.numpy(), [0, 25, 50, 75, 100]).round(1))
print("dur us percentiles  ", np.percentile(dur.numpy(), [0, 25, 50, 75, 100]).round(1))
print("end us max", float((start + dur).max()))
print("prologue (LDS staging) us percentiles", np.percentile(raw[:, 3].double().numpy() / 100.0, [0, 50, 100]).round(2))
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for _ in range(50):
    ops.decode_fwd(grid, blob, lattice=(nx, 1.1, 0, nx ** 3), out=out)
ev1.record(); torch.cuda.synchronize()
print("event ms per launch", ev0.elapsed_time(ev1) / 50)
clk = d[:, 0] / d[:, 1] * 100e6
print("blocks", d.shape[0], "shader cycles/block median %.0f" % d[:, 0].median().item(),
      "real us median %.1f" % (d[:, 1].median().item() / 100), "clock GHz median %.3f min %.3f max %.3f" % (clk.median().item() / 1e9, clk.min().item() / 1e9, clk.max().item() / 1e9))
