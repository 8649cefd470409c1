#!/usr/bin/env python3
"""Golden vectors for the hand encoder's 2-D U-Net (tests/golden/g19_plane_unet.npz) from the REAL reference module
src/encoder/unet.py (UNet(num_classes, in_channels, depth, start_filts), as src/encoder/pointnet.py:49-50 builds it) at the SHIPPED
shape -- depth 4, 32 filters, 32 -> 32 channels, three 32 x 32 planes -- forward and backward:

    out = UNet(x);  L = sum(out * w);  dL/dx and dL/d(every parameter)

The 1.93 M parameters are not stored: the reference module and vtaco_amd.encoder.unet.UNet draw identical parameters from
torch.manual_seed(seed) (same construction order, same initialisers: checked here, and the test re-checks every tensor's float64
sum and abs-sum from the fixture); biases (zero after reset_params) get seeded values.  Stored: x, w, out, dL/dx, and per parameter
the gradient's sum, abs-sum and 64 sampled entries.  Runs only in the build container (/root/reference).

    python tests/golden/make_plane_unet_goldens.py
"""
import importlib.util
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
from make_goldens import _save  # noqa: E402

SEED, BIAS_SEED, DATA_SEED, SAMPLE_SEED = 1900, 1901, 1902, 1903


def seeded_biases(net):
    g = torch.Generator().manual_seed(BIAS_SEED)
    with torch.no_grad():
        for name, p in net.named_parameters():
            if name.endswith("bias"):
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)


def sample_index(name, numel):
    g = torch.Generator().manual_seed(SAMPLE_SEED + sum(map(ord, name)))
    return torch.randint(0, numel, (64,), generator=g)


def main():
    spec = importlib.util.spec_from_file_location("ref_unet", "/root/reference/src/encoder/unet.py")
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    from vtaco_amd.encoder.unet import UNet as Mirror
    torch.set_num_threads(8)
    torch.manual_seed(SEED)
    net = ref.UNet(32, in_channels=32, depth=4, start_filts=32, merge_mode="concat")
    torch.manual_seed(SEED)
    mirror = Mirror(32, in_channels=32, depth=4, start_filts=32, merge_mode="concat")
    sd, md = net.state_dict(), mirror.state_dict()
    assert list(sd) == list(md) and all(torch.equal(sd[k], md[k]) for k in sd), "the mirror does not draw the reference's parameters"
    seeded_biases(net)
    g = torch.Generator().manual_seed(DATA_SEED)
    x = torch.randn(3, 32, 32, 32, generator=g)
    x[x.abs() < 0.4] = 0.0                                          # the planes of a scene are mostly empty cells
    w = torch.randn(3, 32, 32, 32, generator=g)
    x.requires_grad_(True)
    out = net(x)
    (out * w).sum().backward()
    arrs = {"x": x.detach().numpy(), "w": w.numpy(), "out": out.detach().numpy(), "dx": x.grad.numpy(),
            "seeds": np.array([SEED, BIAS_SEED, DATA_SEED, SAMPLE_SEED], dtype=np.int64),
            "names": np.array(list(sd), dtype="U64")}
    for name, p in net.named_parameters():
        gr = p.grad.double().reshape(-1)
        arrs[f"psum.{name}"] = np.array([float(p.detach().double().sum()), float(p.detach().double().abs().sum())])
        arrs[f"gsum.{name}"] = np.array([float(gr.sum()), float(gr.abs().sum())])
        arrs[f"gsample.{name}"] = gr[sample_index(name, gr.numel())].numpy().astype(np.float32)
    _save("g19_plane_unet.npz", **arrs)


if __name__ == "__main__":
    main()
