import sys, time, torch
sys.path.insert(0, "/root/repo")
from vtaco_amd import ops
from vtaco_amd.bench_util import build_scene
dev = torch.device("cuda:0")
sc = build_scene(0, dev)
dec = sc["model"].decoder
B, N = 8, 2048
g = torch.Generator().manual_seed(0)
grid = torch.randn(B, 32, 64, 64, 64, generator=g).to(dev).contiguous(memory_format=torch.channels_last_3d)
p = ((torch.rand(B, N, 3, generator=g) - 0.5) * 1.1).to(dev)
save = ops.decode_save_buffer(B * N, dev)
out = ops.decode_fwd(grid, dec._blob(), pts=p, save=save)
go = torch.randn(B, N, generator=g).to(dev)
bt = dec._blob_t()
def timed(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for want in (True, False):
    print("decode_bwd want_grid_grad", want, timed(lambda: ops.decode_bwd(tuple(grid.shape), bt, go, save, pts=p, want_grid_grad=want)), "ms")
# same points clustered in a small region (atomics collide) and a sorted order
p2 = p.clone(); p2[..., :] *= 0.05
save2 = ops.decode_save_buffer(B * N, dev); ops.decode_fwd(grid, dec._blob(), pts=p2, save=save2)
print("clustered points", timed(lambda: ops.decode_bwd(tuple(grid.shape), bt, go, save2, pts=p2, want_grid_grad=True)), "ms")
print("zeros 268MB", timed(lambda: torch.zeros((B, 64, 64, 64, 32), device=dev)), "ms")
gf = torch.randn(B, N, 32, device=dev)
print("sample_grid_bwd", timed(lambda: ops.sample_grid_bwd(tuple(grid.shape), p, gf)), "ms")
# sorted (by trilinear cell) against per-point scatter: same gradient to rounding, time on uniform and on clustered points
import importlib
for name, pp, sv in (("uniform", p, save), ("clustered", p2, save2)):
    outs = {}
    for mode in (True, False):
        ops.GRID_SCATTER_SORTED = mode
        outs[mode] = ops.decode_bwd(tuple(grid.shape), bt, go, sv, pts=pp, want_grid_grad=True)[0]
        print(name, "sorted" if mode else "per point", timed(lambda: ops.decode_bwd(tuple(grid.shape), bt, go, sv, pts=pp, want_grid_grad=True)), "ms")
    d = (outs[True] - outs[False]).abs().max().item(); m = outs[False].abs().max().item()
    print(name, "max abs difference", d, "of", m)
