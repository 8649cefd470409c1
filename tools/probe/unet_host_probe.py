"""How fast is the host PyTorch-ROCm UNet3D (training path) under different MIOpen settings?"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vtaco_amd.encoder.unet3d import UNet3D
dev = "cuda:0"
def run(tag, bench, cl, B=2):
    torch.backends.cudnn.benchmark = bench
    torch.manual_seed(0)
    net = UNet3D(in_channels=32, out_channels=32, f_maps=32, num_levels=4).to(dev)
    x = torch.randn(B, 32, 64, 64, 64, device=dev)
    if cl:
        net = net.to(memory_format=torch.channels_last_3d); x = x.contiguous(memory_format=torch.channels_last_3d)
    x.requires_grad_(True)
    def step():
        y = net(x); y.sum().backward()
    t0 = time.perf_counter(); step(); torch.cuda.synchronize(); first = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(3): step()
    torch.cuda.synchronize()
    print(f"{tag:40s} first {first:7.2f} s   steady {1e3 * (time.perf_counter() - t0) / 3:9.1f} ms per fwd+bwd (B={B})", flush=True)
run("benchmark off, channels_last_3d, B=2", False, True, 2)
run("benchmark on,  channels_last_3d, B=2", True, True, 2)
run("benchmark off, channels_last_3d, B=8", False, True, 8)
run("benchmark on,  channels_last_3d, B=8", True, True, 8)
