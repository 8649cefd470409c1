"""Which host-PyTorch (MIOpen) layers of the config-4 training step are slow: per-layer forward times of the tactile feature
encoder (Resnet18) and of the t2d depth U-Net on the step's 40 images, with grad mode on and off."""
import sys, time, torch
sys.path.insert(0, '/root/repo')
import torch.nn as nn
from vtaco_amd.bench_util import build_train_case
dev = torch.device("cuda:0")
model, trainer, batch, vf = build_train_case(dev, 0, scenes=8, grad_sync=False)

def hook_times(mod, x, grad):
    recs, hooks = [], []
    def pre(m, inp):
        torch.cuda.synchronize(); m._t0 = time.perf_counter()
    def post(m, inp, out):
        torch.cuda.synchronize(); recs.append((m._name, tuple(inp[0].shape), (time.perf_counter() - m._t0) * 1e3))
    for n, m in mod.named_modules():
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d, nn.MaxPool2d, nn.BatchNorm2d, nn.Linear, nn.Upsample)):
            m._name = n + ":" + m.__class__.__name__ + (str((m.in_channels, m.out_channels, m.kernel_size, m.stride, m.padding)) if hasattr(m, "in_channels") and hasattr(m, "kernel_size") else "")
            hooks.append(m.register_forward_pre_hook(pre)); hooks.append(m.register_forward_hook(post))
    with torch.set_grad_enabled(grad):
        mod(x); recs.clear(); mod(x)
    for h in hooks: h.remove()
    return recs

imgs = batch['inputs.img'].to(dev)
x = imgs.reshape(-1, *imgs.shape[2:]).float()
print("modules:", [(n, type(m).__name__) for n, m in model.named_children()])
t2d = model.encoder_t2d
print("t2d children:", [(n, type(m).__name__) for n, m in t2d.named_children()])
cands = [("encoder_img (Resnet18)", model.encoder_img)] + [("t2d." + n, m) for n, m in t2d.named_children() if sum(1 for _ in m.parameters()) > 0]
for name, mod in cands:
    for grad in (False, True):
        try:
            recs = hook_times(mod, x, grad)
        except Exception as e:
            print("==", name, "grad", grad, "failed:", repr(e)[:160]); continue
        tot = sum(r[2] for r in recs)
        print("==", name, type(mod).__name__, "grad", grad, "sum of layer times %.1f ms" % tot)
        for r in sorted(recs, key=lambda r: -r[2])[:6]: print("  %8.3f ms  %s  in %s" % (r[2], r[0], r[1]))

# the hand encoder's 2-D U-Net on its three 32^2 planes per scene
hand = model.encoder_hand
if getattr(hand, "unet", None) is not None:
    planes = torch.randn(24, hand.c_dim, hand.reso_plane, hand.reso_plane, device=dev)
    for grad in (False, True):
        recs = hook_times(hand.unet, planes, grad)
        print("== encoder_hand.unet", tuple(planes.shape), "grad", grad, "sum of layer times %.1f ms" % sum(r[2] for r in recs))
        for r in sorted(recs, key=lambda r: -r[2])[:8]: print("  %8.3f ms  %s  in %s" % (r[2], r[0], r[1]))
