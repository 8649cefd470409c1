"""How far the split-bf16 UNet3D sits from the exact-f32 UNet3D: grid and decoded logits on the bench scene, and a
few random small nets (the stress test's configurations).  Run with VTACO_CONV_THIN=0/1 to compare tile policies."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vtaco_amd.bench_util import build_scene
from vtaco_amd.encoder.unet3d import UNet3D
dev = torch.device("cuda:0")
sc = build_scene(0, dev)
model, pc = sc["model"], sc["cloud"].to(dev)
enc = model.encoder
out = {}
with torch.no_grad():
    for prec in ("f32", "bf16x3"):
        enc.unet3d.precision = prec
        grid = model.encode_inputs(pc)["grid"]
        out[prec] = (grid.clone(), model.decoder.decode_lattice(grid, 128, precision="f32").clone())
print("bench scene: grid err %.2e (|grid| %.2f), logits err %.2e (|logits| %.2f)" % (
    float((out["f32"][0] - out["bf16x3"][0]).abs().max()), float(out["f32"][0].abs().max()),
    float((out["f32"][1] - out["bf16x3"][1]).abs().max()), float(out["f32"][1].abs().max())))
rng = np.random.RandomState(1)
worst = {}
for it in range(24):
    R, levels = ((16, 2), (16, 3), (32, 3), (32, 4))[it % 4]
    B = 1 + (it // 4) % 2
    torch.manual_seed(int(rng.randint(1 << 30)))
    net = UNet3D(in_channels=32, out_channels=32, f_maps=32, num_levels=levels).to(dev)
    g = torch.Generator().manual_seed(int(rng.randint(1 << 30)))
    x = (torch.randn(B, 32, R, R, R, generator=g) * (torch.rand(B, 1, R, R, R, generator=g) < [0.02, 0.3, 1.0][it % 3])).to(dev)
    xc = x.permute(0, 2, 3, 4, 1).contiguous()
    with torch.no_grad():
        net.precision = "f32"; a = net.forward_channels_last(xc)
        net.precision = "bf16x3"; b = net.forward_channels_last(xc)
    e = float((a - b).abs().max()) / max(1.0, float(a.abs().max()))
    worst[(R, levels, B)] = max(worst.get((R, levels, B), 0.0), e)
for k, v in sorted(worst.items()):
    print("R=%d levels=%d B=%d: worst split-vs-f32 error / max(1,|y|) = %.2e" % (k + (v,)))
