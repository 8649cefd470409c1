"""The tactile Resnet18's share of a training step (8 scenes x five 320x240 images, train-mode BatchNorm, forward + backward) under
torch.backends.cudnn.benchmark = False (MIOpen's immediate mode: what the step runs) and True (MIOpen's find mode), with the time the
first calls take (the find's kernel builds on a box without a MIOpen cache) and the channels-last memory format."""
import sys, time
import torch
sys.path.insert(0, '/root/repo')
from vtaco_amd.encoder import encoder_dict
dev = torch.device('cuda:0')
mode = sys.argv[1] if len(sys.argv) > 1 else "0"
torch.backends.cudnn.benchmark = mode in ("1", "1cl")
torch.manual_seed(0)
net = encoder_dict["Resnet18"](num_classes=32).to(dev).train()
imgs = torch.rand(8, 5, 3, 320, 240, device=dev)
if mode.endswith("cl"):
    net = net.to(memory_format=torch.channels_last)
    imgs = imgs.reshape(40, 3, 320, 240).contiguous(memory_format=torch.channels_last).reshape(8, 5, 3, 320, 240)

def loop():
    net.zero_grad(set_to_none=True)
    out = torch.cat([net(imgs[b]).reshape(1, 5, -1) for b in range(8)])
    out.sum().backward()

t0 = time.perf_counter(); loop(); torch.cuda.synchronize(); first = time.perf_counter() - t0
for _ in range(2):
    loop()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10):
    loop()
torch.cuda.synchronize()
print(f"mode {mode}: first call {first:.1f} s, steady {1e3 * (time.perf_counter() - t0) / 10:.2f} ms per step")
