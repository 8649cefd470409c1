"""GPU debugging aid: decode parity with subsets of the weights zeroed."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import vtaco_oracle as orc
from vtaco_amd import ops

z = np.load("tests/golden/g1_decode.npz")
sd0 = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd.")}
dev = torch.device("cuda:0")

def blob(sd):
    g = lambda k: sd[k].to(dev)
    return ops.pack_decoder(g("fc_p.weight"), g("fc_p.bias"),
        [(g(f"fc_c.{i}.weight"), g(f"fc_c.{i}.bias")) for i in range(5)],
        [(g(f"blocks.{i}.fc_0.weight"), g(f"blocks.{i}.fc_0.bias"), g(f"blocks.{i}.fc_1.weight"), g(f"blocks.{i}.fc_1.bias")) for i in range(5)],
        (g("fc_out.weight"), g("fc_out.bias")))

gen = torch.Generator().manual_seed(0)
grid = torch.randn(1, 32, 8, 8, 8, generator=gen)
pts = (torch.rand(1, 64, 3, generator=gen) - 0.5)

def run(name, keep):
    sd = {k: (v.clone() if any(k.startswith(p) for p in keep) else torch.zeros_like(v)) for k, v in sd0.items()}
    ref = orc.local_decoder_forward(sd, pts, grid)
    got = ops.decode_fwd(grid.to(dev), blob(sd), pts=pts.to(dev)).cpu()
    print(f"{name:40s} maxdiff {float((got-ref).abs().max()):.3e}  ref[:4] {ref[0,:4].tolist()}  got[:4] {got[0,:4].tolist()}")

run("fc_out.bias only", ["fc_out.bias"])
run("fc_p + fc_out", ["fc_p.", "fc_out"])
run("fc_p.bias + fc_out", ["fc_p.bias", "fc_out"])
run("fc_c.0 + fc_out", ["fc_c.0", "fc_out"])
run("fc_c.0.bias + fc_out", ["fc_c.0.bias", "fc_out"])
run("fc_p + fc_c.0 + fc_out", ["fc_p.", "fc_c.0", "fc_out"])
run("fc_p + blocks.0 + fc_out", ["fc_p.", "blocks.0", "fc_out"])
run("fc_p + blocks.0.fc_0+fc1w + fc_out", ["fc_p.", "blocks.0.fc_0", "blocks.0.fc_1.weight", "fc_out"])
run("fc_p + fc_c.1 + fc_out", ["fc_p.", "fc_c.1", "fc_out"])
run("all", [""])
g = torch.randn(2, 32, 5, 6, 7, device=dev)
cl = ops.grid_to_channels_last(g)
print("cl ok", torch.equal(cl, g), torch.equal(ops.grid_from_channels_last(cl), g))
