import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_unet3d_gpu import _unet
DEV = "cuda:0"
R, levels, B, dense = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
net = _unet(32, levels, R + 1).to(DEV)
g = torch.Generator().manual_seed(7)
x = torch.randn(B, 32, R, R, R, generator=g)
if not dense: x = x * (torch.rand(B, 1, R, R, R, generator=g) < 0.1)
x = x.to(DEV)
wgt = torch.randn(B, 32, R, R, R, generator=g).to(DEV)
def rel(a, b): return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-12)
def l2(a, b): return float((a - b).norm()) / max(float(b.norm()), 1e-20)
def grads(xin):
    net.zero_grad(set_to_none=True)
    xg = xin.clone().requires_grad_()
    (net(xg) * wgt).sum().backward()
    return xg.grad.clone(), {n: p.grad.clone() for n, p in net.named_parameters()}
gx0, gp0 = grads(x)
gx1, gp1 = grads(x + 1e-6 * torch.randn_like(x) * (x != 0))
print("host-vs-host(perturbed 1e-6): input", rel(gx1, gx0), "worst param", max(rel(gp1[n], gp0[n]) for n in gp0), "| L2: input %.2e worst param %.2e" % (l2(gx1, gx0), max(l2(gp1[n], gp0[n]) for n in gp0)))
x_cl = x.permute(0, 2, 3, 4, 1).contiguous().requires_grad_()
net.zero_grad(set_to_none=True)
(net.forward_channels_last_train(x_cl) * wgt.permute(0, 2, 3, 4, 1)).sum().backward()
print("hip-vs-host: input", rel(x_cl.grad.permute(0, 4, 1, 2, 3), gx0), "worst param", max(rel(p.grad, gp0[n]) for n, p in net.named_parameters()), "| L2: input %.2e worst param %.2e" % (l2(x_cl.grad.permute(0, 4, 1, 2, 3), gx0), max(l2(p.grad, gp0[n]) for n, p in net.named_parameters())))
import os
if os.environ.get("VTACO_UNET_PRECISION") != "f32":
    net.precision = "f32"
    x2 = x.permute(0, 2, 3, 4, 1).contiguous().requires_grad_()
    net.zero_grad(set_to_none=True)
    (net.forward_channels_last_train(x2) * wgt.permute(0, 2, 3, 4, 1)).sum().backward()
    print("hip(f32)-vs-host: input", rel(x2.grad.permute(0, 4, 1, 2, 3), gx0), "| L2 input %.2e worst param %.2e" % (l2(x2.grad.permute(0, 4, 1, 2, 3), gx0), max(l2(p.grad, gp0[n]) for n, p in net.named_parameters())))
