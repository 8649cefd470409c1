"""How much of the tactile Resnet18's share is the batch size?  Forward + backward over 8 scenes x five 320x240 images as the per-scene
loop (what the reference does: train-mode BatchNorm statistics per scene) against ONE call on the 40 images (plain BatchNorm over all
of them: other statistics -- timing only)."""
import sys, time
import torch
sys.path.insert(0, '/root/repo')
from vtaco_amd.encoder import encoder_dict
dev = torch.device('cuda:0')
torch.manual_seed(0)
net = encoder_dict["Resnet18"](num_classes=32).to(dev).train()
imgs = torch.rand(8, 5, 3, 320, 240, device=dev)

def loop():
    net.zero_grad(set_to_none=True)
    torch.cat([net(imgs[b]).reshape(1, 5, -1) for b in range(8)]).sum().backward()

def one():
    net.zero_grad(set_to_none=True)
    net(imgs.reshape(40, 3, 320, 240)).sum().backward()

for name, fn in (("per-scene loop", loop), ("one call on 40 images", one)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    print(f"{name}: {1e3 * (time.perf_counter() - t0) / 10:.2f} ms per step")
