import sys, torch
sys.path.insert(0, '/root/repo')
from vtaco_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
R = 64
x = torch.randn(1, R, R, R, 32, generator=g).to(dev)
w = (torch.randn(32, 32, 3, 3, 3, generator=g) * 0.05).to(dev)
pf, ps = ops.conv3d_pack(w), ops.conv3d_pack(w, "bf16x3")
for relu in (False, True):
    a, _ = ops.conv3d_gcr(x, None, None, pf, 32, relu, None, want_stats=False)
    b, _ = ops.conv3d_gcr(x, None, None, pf, 32, relu, ps, want_stats=False)
    print("relu", relu, "max rel", float((a - b).abs().max()) / float(a.abs().max()), "l2", float((a - b).norm() / a.norm()))
ref = torch.nn.functional.conv3d(x.permute(0, 4, 1, 2, 3), w, None, padding=1).permute(0, 2, 3, 4, 1)
print("f32 kernel vs torch", float((a.relu() - ref.relu()).abs().max()) / float(ref.abs().max()))
