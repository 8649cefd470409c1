"""Do the VTacOH route's calls agree with each other -- sequential encoders (VTACO_SCENE_OVERLAP=0) and overlapped?  GPU box."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vtaco_amd.bench_util import build_tactile_scene
from vtaco_amd.conv_onet.generation import Generator3D
dev = torch.device("cuda:0")
model, data, depth_origin = build_tactile_scene(dev, variant="vtacoh")
kw = dict(device=dev, resolution0=16, padding=0.1, with_img=True, encode_t2d=False, depth_origin=depth_origin)
first = None
for mode in ("0", "1", "0"):
    os.environ["VTACO_SCENE_OVERLAP"] = mode
    gen = Generator3D(model, **kw)
    for i in range(6):
        np.random.seed(11)
        setup = gen._tactile_setup(data)
        a = setup["anchors"].reshape(-1).double()
        f = setup["feats"].double().sum().item()
        m = gen.generate_obj_mesh_wnf(data)
        key = (m.vertices.shape[0], float(m.vertices.double().sum()), float(a.sum()), f)
        if first is None:
            first = key
        print("overlap", mode, "call", i, "same as first" if key == first else f"DIFFERENT {key} vs {first}")
