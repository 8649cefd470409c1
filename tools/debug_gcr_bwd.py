import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vtaco_amd import ops
from vtaco_amd.encoder.unet3d import _GcrFn, _MaxPoolFn
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
def rel(a, b): return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-12)
for (C1, C2, Cout, R, B) in ((32, 0, 32, 64, 1), (32, 0, 32, 32, 1), (32, 0, 32, 32, 2)):
    x = torch.randn(B, C1, R, R, R, generator=g)
    low = torch.randn(B, C2, R // 2, R // 2, R // 2, generator=g) if C2 else None
    w = torch.randn(Cout, C1 + C2, 3, 3, 3, generator=g) * 0.05
    gamma = 1 + 0.2 * torch.randn(C1 + C2, generator=g); beta = 0.2 * torch.randn(C1 + C2, generator=g)
    wgt = torch.randn(B, Cout, R, R, R, generator=g)
    # reference (CPU autograd)
    xr, wr, gr, br = (t.clone().requires_grad_() for t in (x, w, gamma, beta))
    lr = low.clone().requires_grad_() if C2 else None
    xin = torch.cat([xr, F.interpolate(lr, scale_factor=2, mode="nearest")], 1) if C2 else xr
    y = F.relu(F.conv3d(F.group_norm(xin, 8, gr, br, 1e-5), wr, None, padding=1))
    (y * wgt).sum().backward()
    # HIP
    cl = lambda t: t.to(dev).permute(0, 2, 3, 4, 1).contiguous()
    xh = cl(x).requires_grad_(); lh = cl(low).requires_grad_() if C2 else None
    wh, gh, bh = (t.to(dev).requires_grad_() for t in (w, gamma, beta))
    xp = ops.channel_stats(xh.detach())[0]; lp = ops.channel_stats(lh.detach())[0] if C2 else None
    yh, _ = _GcrFn.apply(xh, lh, gh, bh, wh, xp, lp, 8, 1e-5, os.environ.get("PREC", "f32"))
    print("fwd", rel(yh.permute(0, 4, 1, 2, 3).detach().cpu(), y.detach()))
    (yh * cl(wgt)).sum().backward()
    print((C1, C2, Cout, R), "dx", rel(xh.grad.permute(0, 4, 1, 2, 3).cpu(), xr.grad), "dw", rel(wh.grad.cpu(), wr.grad),
          "dgamma", rel(gh.grad.cpu(), gr.grad), "dbeta", rel(bh.grad.cpu(), br.grad),
          "dlow", rel(lh.grad.permute(0, 4, 1, 2, 3).cpu(), lr.grad) if C2 else None)
x = torch.randn(1, 32, 8, 8, 8, generator=g).relu()
xr = x.clone().requires_grad_(); wgt = torch.randn(1, 32, 4, 4, 4, generator=g)
(F.max_pool3d(xr, 2) * wgt).sum().backward()
xh = x.to(dev).permute(0, 2, 3, 4, 1).contiguous().requires_grad_()
(_MaxPoolFn.apply(xh) * wgt.to(dev).permute(0, 2, 3, 4, 1)).sum().backward()
print("maxpool dx", rel(xh.grad.permute(0, 4, 1, 2, 3).cpu(), xr.grad))
