import os, sys, torch
sys.path.insert(0, '/root/repo')
from vtaco_amd import ops
dev = torch.device("cuda:0")
R, C1, Cout = 64, 32, 32
g = torch.Generator().manual_seed(1)
def run(x, w, tag):
    gamma, beta = torch.ones(C1, device=dev), torch.zeros(C1, device=dev)
    xs = ops.channel_stats(x)
    pf, ph = ops.conv3d_pack(w), ops.conv3d_pack(w, precision="f16x3")
    ss = ops.gn_scale_shift(xs, None, C1, 0, 1, R ** 3, gamma, beta, 8, 1e-5, dev)
    fn = lambda: ops.conv3d_gcr(x, None, ss, pf, Cout, True, None, packed_w_f16x3=ph)
    for _ in range(200): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): fn()
    e1.record(); torch.cuda.synchronize()
    print(tag, "%.1f us per conv" % (e0.elapsed_time(e1) / 200 * 1e3))
x = torch.randn(1, R, R, R, C1, generator=g).to(dev)
w = (torch.randn(Cout, C1, 3, 3, 3, generator=g) * 0.05).to(dev)
run(x, w, "random data ")
run(torch.zeros_like(x), torch.zeros_like(w), "all-zero data")
run(x, torch.zeros_like(w), "zero weights ")
