import sys, torch
sys.path.insert(0, "/root/repo")
from vtaco_amd import ops
from vtaco_amd.common import make_3d_grid
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
for nx, R in ((64, 32), (128, 64), (32, 16)):
    grid = ops.grid_to_channels_last(torch.randn(1, 32, R, R, R, generator=g).to(dev))
    pts = (1.1 * make_3d_grid((-0.5,) * 3, (0.5,) * 3, (nx,) * 3)).to(dev)
    want = ops.sample_grid(grid, pts.unsqueeze(0))
    got = ops.sample_grid(grid, None, lattice=(nx, 1.1, 0, nx ** 3))
    import os
    d = (got - want).abs()
    bad = torch.nonzero(d.amax(dim=2)[0] > 0).flatten()
    print(nx, R, "max diff", float(d.max()), "bad points", bad.numel(), "of", nx ** 3)
    if bad.numel():
        i = int(bad[0]); print(" first bad", i, (i // (nx * nx), (i // nx) % nx, i % nx), got[0, i, :4].tolist(), want[0, i, :4].tolist())
        ix = bad // (nx * nx); iy = (bad // nx) % nx; iz = bad % nx
        print(" bad ix", sorted(set(ix.tolist()))[:20], "iy", sorted(set(iy.tolist()))[:20], "iz", sorted(set(iz.tolist()))[:20])
