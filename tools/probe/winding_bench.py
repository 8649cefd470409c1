import sys, time, torch
sys.path.insert(0, '/root/repo')
from vtaco_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for F in (10000, 50000):
    verts = torch.randn(F // 2 + 2, 3, generator=g).to(dev)
    faces = torch.randint(0, F // 2 + 2, (F, 3), generator=g).to(dev)
    pts = torch.randn(16384, 3, generator=g).to(dev)
    for _ in range(2): ops.winding_number(verts, faces, pts)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): ops.winding_number(verts, faces, pts)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"F={F}: {dt*1e3:.2f} ms for 16384 points = {16384*F/dt/1e9:.1f} G solid angles/s")
