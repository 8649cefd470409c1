#!/usr/bin/env python3
"""Golden vectors for the VTacO (t2d) branch of ``Generator3D.generate_obj_mesh_wnf`` (generation.py:202-257) from the REAL
reference: which lattice points receive which finger's tactile feature, given the sample's depth images, camera poses and
the dataset's ``depth_origin`` (g12_t2d.npz).

Build container only.  The reference generator runs on a stand-in model (fixed tensors; its tactile features are the
constant rows t+1, so the dense c_img_all it hands to eval_points reads back as a finger id per lattice point) and stops
at eval_points.  generation.py imports trimesh / skimage (not installed) and reads ./data/VTacO_mesh/depth_origin.txt at
import: stand-ins for the modules, and np.loadtxt answers that one read with THIS script's synthetic depth_origin.

    python tests/golden/make_t2d_goldens.py
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_goldens as mg          # noqa: E402

W, H = 240, 320                    # generation.py:18-19 (w, h)


class _Stop(Exception):
    pass


def main():
    mg._install_stubs()
    rs = np.random.RandomState(50)
    # the flat sensor reading, and five depth images that dent it where the gel touches (a blob per finger)
    depth_origin = (0.0215 + 1e-5 * rs.randn(W * H)).astype(np.float64)
    depths = np.repeat(depth_origin[None, :].astype(np.float32), 5, axis=0)
    yy, xx = np.mgrid[0:H, 0:W]
    for t, (cy, cx, rad) in enumerate([(160, 120, 30), (80, 60, 12), (250, 180, 20), (40, 200, 6), (300, 30, 25)]):
        blob = ((yy - cy) ** 2 + (xx - cx) ** 2) < rad ** 2
        depths[t, blob.reshape(-1)] -= (0.0004 + 0.001 * rs.rand(int(blob.sum()))).astype(np.float32)
    for name in ("trimesh", "skimage", "skimage.measure"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["skimage"].measure = sys.modules["skimage.measure"]
    loadtxt = np.loadtxt
    np.loadtxt = lambda *a, **k: depth_origin.copy()
    try:
        generation = importlib.import_module("src.conv_onet.generation")
    finally:
        np.loadtxt = loadtxt

    g = torch.Generator().manual_seed(51)
    pc_ply = torch.randn(1, 500, 3, generator=g) * 0.15 + 0.02
    cam_pos = torch.randn(1, 5, 3, generator=g) * 0.12
    cam_rot = torch.randn(1, 5, 3, generator=g) * 0.8
    touch = torch.tensor([[True, True, True, False, True]])
    c_img = (torch.arange(5).float() + 1).view(1, 5, 1).expand(1, 5, 32).contiguous()
    data = {"inputs": torch.zeros(1, 16, 3), "inputs.img": torch.zeros(1, 5, 3, 8, 6), "inputs.depth": torch.from_numpy(depths)[None],
            "inputs.touch_success": touch, "inputs.pc_ply": pc_ply, "points.mano": torch.zeros(1, 51),
            "points.points_obj": torch.zeros(1, 8, 3), "points.wrist": torch.zeros(1, 3), "points.cam_pos": cam_pos, "points.cam_rot": cam_rot}
    seen = {}

    class FakeModel(object):
        def to(self, device):
            return self

        def eval(self):
            return self

        def encode_t2d(self, inputs, imgs):
            return torch.zeros(1, 5, W * H), {"mano_param": torch.zeros(1, 30)}

        def encode_inputs(self, inputs):
            return "c"

        def encode_hand_inputs(self, inputs):
            return {}

        def encode_img_inputs(self, imgs):
            return c_img

    gen = generation.Generator3D(FakeModel(), device="cpu", resolution0=32, padding=0.1, with_img=True, encode_t2d=True)

    def eval_points(p, c=None, c_img_all=None, **kw):
        seen["ids"] = c_img_all[0, :, 0].round().to(torch.uint8).clone()
        raise _Stop()

    gen.eval_points = eval_points
    np.random.seed(321)
    try:
        gen.generate_obj_mesh_wnf(data)
    except _Stop:
        pass
    ids = seen["ids"].numpy()
    ids = np.where(ids == 0, 255, ids - 1).astype(np.uint8)            # 255 = no tactile feature, else the finger
    print("lattice points per finger:", [int((ids == f).sum()) for f in range(5)])
    mg._save("g12_t2d.npz", depth_origin=depth_origin, depths=depths, pc_ply=pc_ply.numpy(), cam_pos=cam_pos.numpy(),
             cam_rot=cam_rot.numpy(), touch=touch.numpy(), ids=ids, seed=np.array(321), nx=np.array(128))


if __name__ == "__main__":
    main()
