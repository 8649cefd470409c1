"""Weight-gradient kernels of the UNet3D backward at the training shapes (8 scenes): f32 MFMA kernel vs the split-half one."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vtaco_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
SHAPES = ((8, 64, 32, 0, 32), (8, 64, 32, 64, 32), (8, 32, 64, 0, 64), (8, 32, 64, 128, 64), (8, 16, 128, 256, 128), (8, 8, 128, 0, 256))
for B, R, C1, C2, Cout in (SHAPES[:1] if "--first" in sys.argv else SHAPES):
    x = torch.randn(B, R, R, R, C1, generator=g).to(dev)
    low = torch.randn(B, R // 2, R // 2, R // 2, C2, generator=g).to(dev) if C2 else None
    gr = (torch.randn(B, R, R, R, Cout, generator=g) * 1e-5).to(dev)
    ss = torch.stack((torch.ones(B, C1 + C2), torch.zeros(B, C1 + C2)), -1).to(dev).contiguous()
    gmax = gr.abs().max().reshape(1)
    res = []
    for prec in ("f32", "f16x3"):
        for _ in range(3):
            ops.conv3d_wgrad(x, low, ss, gr, precision=prec, g_absmax=gmax)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            ops.conv3d_wgrad(x, low, ss, gr, precision=prec, g_absmax=gmax)
        b.record(); b.synchronize()
        res.append(a.elapsed_time(b) / 10)
    gf = 2.0 * B * R ** 3 * 27 * (C1 + C2) * Cout / 1e9
    print(f"B={B} R={R} {C1}+{C2}->{Cout}: f32 {res[0]:.3f} ms ({gf / res[0]:.0f} TFLOP/s)  f16x3 {res[1]:.3f} ms ({gf / res[1]:.0f} TFLOP/s algorithmic)", flush=True)
