"""Time of the hand encoder's 2-D U-Net as the one-launch HIP kernel against the nn.Conv2d modules (MIOpen), per shape; and of the whole
hand encoder (eager and as a graph replay)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vtaco_amd import ops
from vtaco_amd.encoder.unet import UNet
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True


def ev_ms(fn, n=200, warm=20):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for depth, cin, start, classes, n_img, H, W in ((4, 32, 32, 32, 3, 32, 32), (4, 32, 32, 32, 24, 32, 32), (4, 512, 32, 512, 3, 64, 64), (4, 512, 32, 512, 24, 64, 64)):
    torch.manual_seed(0)
    net = UNet(classes, in_channels=cin, depth=depth, start_filts=start).to(dev).eval()
    x = torch.randn(n_img, cin, H, W, device=dev)
    with torch.no_grad():
        t_hip = ev_ms(lambda: net(x))
        t_mod = ev_ms(lambda: net.forward_modules(x), 50, 5)
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            net(x)
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            y = net(x)
        t_graph = ev_ms(g.replay)
    print(f"U-Net depth {depth} {cin}->{classes} start {start}, {n_img} x {H}x{W}: HIP one launch {t_hip * 1e3:.1f} us (graph replay {t_graph * 1e3:.1f} us), "
          f"nn.Conv2d modules {t_mod * 1e3:.1f} us")
