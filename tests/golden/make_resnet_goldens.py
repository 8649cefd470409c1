#!/usr/bin/env python3
"""Golden vector for the tactile feature encoder (g14_resnet.npz) from the REAL reference ``Resnet18`` (src/layers.py:127-195).
Build container only.  The 11 M parameters are not stored: both sides fill them with ``deterministic_fill`` (seeded values in
state_dict order), so the fixture holds an input, the outputs and the list of parameter names / shapes.

    python tests/golden/make_resnet_goldens.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_goldens as mg          # noqa: E402


def deterministic_fill(module, seed):
    """Overwrite every parameter and buffer, in state_dict order, with seeded values (BatchNorm variances positive): two modules
    with the same keys and shapes get the same numbers without storing them."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, t in module.state_dict().items():
            if "num_batches_tracked" in name:
                continue
            r = torch.randn(t.shape, generator=g)
            if name.endswith("running_var"):
                t.copy_(0.5 + r.abs())
            elif name.endswith("bn1.weight") or name.endswith("bn2.weight") or name.endswith(".1.weight"):
                t.copy_(1.0 + 0.1 * r)
            elif t.dim() >= 2:
                t.copy_(r * (1.0 / max(1, t[0].numel())) ** 0.5)
            else:
                t.copy_(0.1 * r)


def make_resnet_golden():
    """g14_resnet.npz: the reference Resnet18 (tactile feature encoder, src/layers.py:127-195) on a seeded input, its parameters
    filled by ``deterministic_fill`` (11 M numbers that are rebuilt, not stored)."""
    mg._install_stubs()
    from src.layers import Resnet18
    net = Resnet18(num_classes=32)
    deterministic_fill(net, 90)
    g = torch.Generator().manual_seed(91)
    x = torch.rand(2, 3, 96, 64, generator=g)
    net.eval()
    with torch.no_grad():
        y_eval = net(x)
    net.train()
    with torch.no_grad():
        y_train = net(x)
    keys = [f"{k}:{tuple(v.shape)}" for k, v in net.state_dict().items() if "num_batches_tracked" not in k]
    mg._save("g14_resnet.npz", x=x.numpy(), y_eval=y_eval.numpy(), y_train=y_train.numpy(), keys=np.array(keys))



if __name__ == "__main__":
    make_resnet_golden()
