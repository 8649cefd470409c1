"""Time one 'gcr' conv (GroupNorm -> 3x3x3 conv -> ReLU) on a 64^3 channels-last volume through the C ABI:
exact-f32 kernel vs split-bf16 kernel.  bench_conv.py [C1 C2 Cout]"""
import os, sys, json, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vtaco_amd import ops
dev = torch.device("cuda:0")
C1, C2, Cout = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (32, 0, 32)
R = 64
g = torch.Generator().manual_seed(0)
x = torch.randn(1, R, R, R, C1, generator=g).to(dev)
low = torch.randn(1, R // 2, R // 2, R // 2, C2, generator=g).to(dev) if C2 else None
w = (torch.randn(Cout, C1 + C2, 3, 3, 3, generator=g) * 0.05).to(dev)
gamma, beta = torch.ones(C1 + C2, device=dev), torch.zeros(C1 + C2, device=dev)
xs = ops.channel_stats(x); ls = ops.channel_stats(low) if C2 else None
pf, ps = ops.conv3d_pack(w), ops.conv3d_pack(w, precision="bf16x3")
def run(split):
    return ops.gn_conv3d_relu(x, xs, low, ls, gamma, beta, 8, pf, Cout, packed_w_bf16x3=ps if split else None)
res = {}
for name, split in (("f32", False), ("bf16x3", True)):
    for _ in range(3): run(split)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): run(split)
    torch.cuda.synchronize(); res[name + "_us"] = (time.perf_counter() - t0) / 20 * 1e6
res["gflop"] = 2 * 27 * (C1 + C2) * Cout * R ** 3 / 1e9
print(json.dumps(res))
