"""The host PyTorch-ROCm modules that sit ON the path by north_star's choice (tactile U-Net depth estimator, Resnet18 tactile
feature encoder: reference src/layers.py:322-450, 54-207) run through MIOpen / rocBLAS on the MI355X: their results are pinned
here against the same reference-made goldens the CPU tests use (g6_tactile.npz, g14_resnet.npz), so that library drift on the
device is measured, not assumed.  (They are plumbing, not product kernels; the test still belongs to the GPU suite because the
numbers that feed the HIP decoder come from the device run.)"""
import json
import os
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
T = torch.from_numpy


def _report(name, rep):
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        path = os.path.join(out, "host_modules_gpu.json")
        cur = json.load(open(path)) if os.path.exists(path) else {}
        cur[name] = rep
        json.dump(cur, open(path, "w"), indent=1, sort_keys=True)
    except (OSError, ValueError):
        pass


def test_tactile_unet_on_the_device_eval_and_train_bn():
    from vtaco_amd.encoder import encoder_dict
    a, sd = load_golden("g6_tactile.npz")
    net = encoder_dict["UNet"](num_classes=1, in_channels=3, depth=3, start_filts=8)
    net.load_state_dict(sd, strict=False)
    net = net.to(DEV)
    x = T(a["x"]).to(DEV)
    rep = {}
    for mode, key in (("eval", "y_eval"), ("train", "y_train")):
        net.train(mode == "train")
        with torch.no_grad():
            y = net(x).cpu()
        ref = T(a[key])
        rep[mode] = {"max_abs_err": float((y - ref).abs().max()), "output_max": float(ref.abs().max())}
        assert rep[mode]["max_abs_err"] <= 1e-4 * max(1.0, rep[mode]["output_max"]), rep
    _report("tactile_unet", rep)


def test_resnet18_tactile_features_on_the_device():
    sys.path.insert(0, GOLDEN)
    from make_resnet_goldens import deterministic_fill
    from vtaco_amd.encoder import encoder_dict
    z = np.load(os.path.join(GOLDEN, "g14_resnet.npz"))
    net = encoder_dict["Resnet18"](num_classes=32)
    deterministic_fill(net, 90)
    net = net.to(DEV)
    x = T(z["x"]).to(DEV)
    rep = {}
    for mode, key in (("eval", "y_eval"), ("train", "y_train")):
        net.train(mode == "train")
        with torch.no_grad():
            y = net(x).cpu()
        ref = T(z[key])
        rep[mode] = {"max_abs_err": float((y - ref).abs().max()), "output_max": float(ref.abs().max())}
        assert rep[mode]["max_abs_err"] <= 1e-4 * rep[mode]["output_max"], rep
    _report("resnet18", rep)
