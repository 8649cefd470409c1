import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """Load tests/golden/<name>; returns (arrays, state_dicts) where keys
    'sd.<x>' are gathered into a {x: tensor} dict."""
    z = np.load(os.path.join(GOLDEN, name))
    arrs, sd = {}, {}
    for k in z.files:
        v = z[k]
        if k.startswith("sd."):
            sd[k[3:]] = torch.from_numpy(v)
        else:
            arrs[k] = v
    return arrs, sd


def sub_sd(sd, prefix):
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


@pytest.fixture(scope="session")
def golden():
    return load_golden
