#!/usr/bin/env python3
"""Golden vectors for LocalPoolPointnet with scatter_type='mean' (tests/golden/g18_pointnet_mean.npz), from the REAL reference:
src/encoder/pointnet.py:32-166 (pool_local with torch_scatter.scatter_mean, :64-69 and :116-132) on the object grid and on the hand
encoder's three planes -- the first and last block's outputs, fc_c and the scatter-mean grid / planes of a seeded two-scene cloud with outliers that hit
both clamps.  Runs only in the build container (/root/reference); torch_scatter==2.0.9 is not installed: make_goldens.py's stand-in
(the two ops written with torch primitives).

    python tests/golden/make_pointnet_mean_goldens.py
"""
import importlib
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_goldens import _install_stubs, _randomise, _save, _sd  # noqa: E402


def main():
    _install_stubs()
    pointnet = importlib.import_module("src.encoder.pointnet")
    torch.set_num_threads(8)
    out = {}
    g = torch.Generator().manual_seed(181)
    d = torch.randn(2, 1500, 3, generator=g)
    p_in = 0.3 * d / d.norm(dim=-1, keepdim=True) + 0.005 * torch.randn(2, 1500, 3, generator=g)
    p_in[:, :30] = (torch.rand(2, 30, 3, generator=g) - 0.5) * 1.4      # outliers -> clamps
    p_in[1, 30:330] = p_in[1, 30:31]                                      # 300 points in one cell: a long segment
    out["p"] = p_in.numpy()
    for tag, kw in (("grid", dict(grid_resolution=16, plane_type="grid")),
                    ("planes", dict(plane_resolution=16, plane_type=["xz", "xy", "yz"]))):
        torch.manual_seed(182)
        enc = pointnet.LocalPoolPointnet(c_dim=32, dim=3, hidden_dim=32, scatter_type="mean", unet3d=False, unet=False,
                                         padding=0.1, n_blocks=5, **kw)
        _randomise(enc, 183)
        stages, fcc = [], []
        hooks = [blk.register_forward_hook(lambda m, i, o: stages.append(o.detach().clone())) for blk in enc.blocks]
        hooks.append(enc.fc_c.register_forward_hook(lambda m, i, o: fcc.append(o.detach().clone())))
        with torch.no_grad():
            fea = enc(p_in)
        for h in hooks:
            h.remove()
        out[f"{tag}.stage0"], out[f"{tag}.stage4"] = stages[0].numpy(), stages[4].numpy()      # in front of the first pool, behind the last
        out[f"{tag}.fc_c"] = fcc[0].numpy()
        for k, v in fea.items():
            out[f"{tag}.fea.{k}"] = v.numpy().astype(np.float32)
        out.update(_sd(enc, f"sd.{tag}."))
    _save("g18_pointnet_mean.npz", **out)


if __name__ == "__main__":
    main()
