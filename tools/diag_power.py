"""Is the lattice decode clock-limited by power?  Same kernel, same launch, on (a) the bench scene and (b) all-zero
grid + weights (an MFMA on zero operands draws far less power): equal cycle counts with a shorter wall time on zeros
means the chip is holding its clock down under the real data.  Prints ms per launch per precision."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vtaco_amd.bench_util import build_scene
dev = torch.device("cuda:0")
sc = build_scene(0, dev)
dec, grid = sc["model"].decoder, sc["grid"]
nx = 128

def timed(fn, n=300):
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

for prec in ("bf16x3", "f16x3", "f32"):
    real = timed(lambda: dec.decode_lattice(grid, nx, precision=prec))
    with torch.no_grad():
        saved = [p.detach().clone() for p in dec.parameters()]
        for p in dec.parameters():
            p.zero_()
    zgrid = torch.zeros_like(grid)
    zero = timed(lambda: dec.decode_lattice(zgrid, nx, precision=prec))
    with torch.no_grad():
        for p, s in zip(dec.parameters(), saved):
            p.copy_(s)
    print(f"{prec}: real data {real:.4f} ms, all-zero data {zero:.4f} ms, ratio {real / zero:.3f}")
