"""Backward (data + weight gradient) time of every distinct Conv2d configuration of the host-PyTorch nets of the config-4 step
(Resnet18 tactile encoder on 40 images 320x240; the hand encoder's 2-D U-Net on 24 planes 32x32), through MIOpen in its default
(immediate) mode: finds the configurations that fall back to MIOpen's naive kernels."""
import sys, time, torch
sys.path.insert(0, '/root/repo')
import torch.nn as nn
from vtaco_amd.bench_util import build_train_case
dev = torch.device("cuda:0")
model, trainer, batch, vf = build_train_case(dev, 0, scenes=8, grad_sync=False)

def shapes_of(mod, x):
    recs, hooks = {}, []
    def post(m, inp, out):
        recs[(m.in_channels, m.out_channels, m.kernel_size, m.stride, m.padding, type(m).__name__, tuple(inp[0].shape))] = m
    for n, m in mod.named_modules():
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
            hooks.append(m.register_forward_hook(post))
    with torch.no_grad():
        mod(x)
    for h in hooks: h.remove()
    return recs

def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

imgs = batch['inputs.img'].to(dev)
x = imgs.reshape(-1, *imgs.shape[2:]).float()
hand = model.encoder_hand
cases = [("Resnet18", model.encoder_img, x)]
if getattr(hand, "unet", None) is not None:
    cases.append(("hand U-Net", hand.unet, torch.randn(24, hand.c_dim, hand.reso_plane, hand.reso_plane, device=dev)))
for fmt_name, fmt in (("contiguous", torch.contiguous_format), ("channels_last", torch.channels_last)):
    for name, mod, inp in cases:
        rows = []
        for key, m in shapes_of(mod, inp).items():
            xin = torch.randn(key[-1], device=dev).contiguous(memory_format=fmt).requires_grad_()
            mm = type(m)(m.in_channels, m.out_channels, m.kernel_size, m.stride, m.padding, bias=m.bias is not None).to(dev).to(memory_format=fmt)
            y = mm(xin)
            gy = torch.randn_like(y)
            fwd = timed(lambda: mm(xin))
            def bwd():
                yy = mm(xin)
                torch.autograd.grad(yy, [xin, mm.weight], gy)
            tot = timed(bwd)
            rows.append((tot - fwd, fwd, key))
        print("==", name, fmt_name, "backward total %.1f ms, forward %.1f ms" % (sum(r[0] for r in rows), sum(r[1] for r in rows)))
        for r in sorted(rows, key=lambda r: -r[0])[:5]:
            print("   bwd %8.3f ms  fwd %7.3f ms  %s" % r)
