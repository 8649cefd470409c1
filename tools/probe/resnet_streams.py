"""Does running the tactile Resnet18's per-scene calls (five 320x240 images each, train-mode BatchNorm) on separate streams pay?
Forward + backward of encode_img_inputs on 8 scenes: the sequential loop against one stream per scene (timing only: the running
statistics race in this probe)."""
import sys, time
import torch
sys.path.insert(0, '/root/repo')
from vtaco_amd.encoder import encoder_dict
dev = torch.device('cuda:0')
torch.manual_seed(0)
net = encoder_dict["Resnet18"](num_classes=32).to(dev).train()
imgs = torch.rand(8, 5, 3, 320, 240, device=dev)

def seq():
    net.zero_grad(set_to_none=True)
    out = torch.cat([net(imgs[b]).reshape(1, 5, -1) for b in range(8)])
    out.sum().backward()

streams = [torch.cuda.Stream() for _ in range(8)]
def par(ns):
    net.zero_grad(set_to_none=True)
    cur = torch.cuda.current_stream()
    outs = [None] * 8
    for b in range(8):
        s = streams[b % ns]
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            outs[b] = net(imgs[b]).reshape(1, 5, -1)
    for s in streams[:ns]:
        cur.wait_stream(s)
    torch.cat(outs).sum().backward()

def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n

print(f"sequential: {timed(seq):.2f} ms")
for ns in (2, 4, 8):
    print(f"{ns} streams: {timed(lambda: par(ns)):.2f} ms")
