#!/usr/bin/env python3
"""Golden vectors for the VTacO (t2d) training-sample assembly (g13_trainer_t2d.npz) from the REAL reference:
``Trainer.compute_loss_t2d_img`` (src/conv_onet/training.py:757-894) on a stand-in model that returns seeded tensors and
records what the trainer hands to ``decode_img``.

Build container only.  Stand-ins: trimesh (never reached), and **igl**: the reference labels its re-sampled points with
``igl.fast_winding_number_for_meshes``, libigl's approximation of the generalized winding number; libigl is not installed, so
the stand-in returns the exact winding number (oracle.winding_number) -- this fixture pins everything around that call
(contact clouds, numpy draws, the ones-filled feature rows, the three losses), not libigl's approximation error.
np.loadtxt answers the module's one dataset read (depth_origin) with the synthetic array of g12_t2d.npz.

    python tests/golden/make_t2d_trainer_goldens.py
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import make_goldens as mg          # noqa: E402
from oracle import vtaco_oracle as orc   # noqa: E402

W, H = 240, 320


def meshes():
    cube_v = (np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0, 0, 1], [1, 0, 1], [1, 1, 1], [0, 1, 1]], dtype=np.float32) - 0.5) * 0.6
    cube_f = np.array([[0, 2, 1], [0, 3, 2], [4, 5, 6], [4, 6, 7], [0, 1, 5], [0, 5, 4], [1, 2, 6], [1, 6, 5], [2, 3, 7], [2, 7, 6],
                       [3, 0, 4], [3, 4, 7]], dtype=np.int64)
    tet_v = np.array([[0.4, 0.4, 0.4], [-0.4, -0.4, 0.4], [-0.4, 0.4, -0.4], [0.4, -0.4, -0.4]], dtype=np.float32)
    tet_f = np.array([[0, 1, 2], [0, 3, 1], [0, 2, 3], [1, 3, 2]], dtype=np.int64)
    return {"cube": {"v": cube_v, "f": cube_f}, "tet": {"v": tet_v, "f": tet_f}}


def main():
    mg._install_stubs()
    z12 = np.load(os.path.join(HERE, "g12_t2d.npz"))
    depth_origin = z12["depth_origin"]
    igl = types.ModuleType("igl")
    igl.fast_winding_number_for_meshes = lambda v, f, q: orc.winding_number(v, f, q)
    sys.modules["igl"] = igl
    sys.modules.setdefault("trimesh", types.ModuleType("trimesh"))
    loadtxt = np.loadtxt
    np.loadtxt = lambda *a, **k: depth_origin.copy()
    try:
        training = importlib.import_module("src.conv_onet.training")
    finally:
        np.loadtxt = loadtxt
    B, N, NS = 2, 1500, 700
    g = torch.Generator().manual_seed(60)
    depths = torch.from_numpy(np.stack([z12["depths"], np.roll(z12["depths"], 7, axis=0)]))          # scene 1: the images permuted
    pc_ply = torch.randn(B, 500, 3, generator=g) * 0.15 + 0.02
    cam_pos = torch.randn(B, 5, 3, generator=g) * 0.12
    cam_rot = torch.randn(B, 5, 3, generator=g) * 0.8
    touch = torch.tensor([[True, True, True, False, True], [False, True, True, True, True]])
    p = (torch.rand(B, N, 3, generator=g) - 0.5) * 1.1
    hand = {"mano_param": torch.randn(B, 51, generator=g) * 0.2, "mano_verts": torch.randn(B, 778, 3, generator=g) * 0.05}
    c_img = torch.randn(B, 5, 32, generator=g)
    # predicted depth of the stand-in: a ramp (rebuilt by the tests, not stored); the depth images are g12's, scene 1 = rolled by 7
    pred_depth = (torch.linspace(0, 1, W * H).view(1, 1, -1) * torch.tensor([0.2, 0.4, 0.6, 0.8, 1.0]).view(1, 5, 1)).expand(B, 5, W * H).contiguous()
    digit = torch.randn(B, 30, generator=g) * 0.3
    mano_gt = torch.randn(B, 51, generator=g) * 0.2
    pc_hand = torch.randn(B, 778, 3, generator=g) * 0.05
    data = {"points": p, "points.occ": torch.zeros(B, N), "points.mano": mano_gt, "points.pc_hand": pc_hand, "points.wrist": torch.zeros(B, 3),
            "points.name": ["cube", "tet"], "points.cam_pos": cam_pos, "points.cam_rot": cam_rot,
            "inputs": torch.zeros(B, 16, 3), "inputs.pc_ply": pc_ply, "inputs.img": torch.zeros(B, 5, 3, 8, 6),
            "inputs.depth": depths, "inputs.touch_success": touch}
    seen = {}

    class FakeModel(object):
        def encode_t2d(self, inputs, imgs):
            return pred_depth, {"mano_param": digit}

        def encode_inputs(self, inputs):
            return "c"

        def encode_hand_inputs(self, inputs):
            return hand

        def encode_img_inputs(self, imgs):
            return c_img

        def decode(self, p_sample, c, **kw):
            seen["p_sample_plain"] = p_sample.detach().clone()
            return types.SimpleNamespace(logits=p_sample.sum(-1) * 0.5)

        def decode_img(self, p_sample, c, c_img_all, **kw):
            seen["p_sample"], seen["c_img_all"] = p_sample.detach().clone(), c_img_all.detach().clone()
            return types.SimpleNamespace(logits=p_sample.sum(-1) * 0.5 + c_img_all.sum(-1) * 0.01)

    l1 = training.F.l1_loss
    calls = []

    def spy_l1(a, b, *args, **kw):
        calls.append(b.detach().clone())
        return l1(a, b, *args, **kw)

    to = torch.Tensor.to
    losses = {}
    for pretrained in (True, False):
        trainer = training.Trainer(FakeModel(), None, device="cpu", num_sample=NS, with_img=True, encode_t2d=True, pretrained_t2d=pretrained)
        np.random.seed(77)
        training.F.l1_loss = spy_l1
        torch.Tensor.to = lambda self, *a, **k: (to(self, *a, **k).clone() if self.is_leaf and self.requires_grad else to(self, *a, **k))
        calls.clear()
        try:
            out = trainer.compute_loss_t2d_img(data, meshes())
        finally:
            training.F.l1_loss = l1
            torch.Tensor.to = to
        losses[pretrained] = [float(x) for x in out]
        if pretrained:
            seen["occ_new"] = calls[0]
    # the variant without tactile features (compute_loss_t2d, training.py:628-755): same assembly, plain decode
    plain = {}
    for pretrained in (True, False):
        trainer = training.Trainer(FakeModel(), None, device="cpu", num_sample=NS, with_img=False, encode_t2d=True, pretrained_t2d=pretrained)
        np.random.seed(77)
        torch.Tensor.to = lambda self, *a, **k: (to(self, *a, **k).clone() if self.is_leaf and self.requires_grad else to(self, *a, **k))
        try:
            plain[pretrained] = [float(x) for x in trainer.compute_loss_t2d(data, meshes())]
        finally:
            torch.Tensor.to = to
    # (that variant normalises the depth images to [0,1] BEFORE looking for contact pixels -- training.py:643-644 -- so nearly every
    # pixel "touches" and its samples differ from the with_img variant's)
    ones_rows = (seen["c_img_all"] == 1).all(-1).sum(1)
    print("rows without a tactile feature (ones):", ones_rows.tolist(), "losses", losses)
    m = meshes()
    mg._save("g13_trainer_t2d.npz", p=p.numpy(), pc_ply=pc_ply.numpy(), cam_pos=cam_pos.numpy(), cam_rot=cam_rot.numpy(),
             touch=touch.numpy(), c_img=c_img.numpy(), digit=digit.numpy(),
             mano=mano_gt.numpy(), pc_hand=pc_hand.numpy(), mano_param=hand["mano_param"].numpy(), mano_verts=hand["mano_verts"].numpy(),
             p_sample=seen["p_sample"].numpy(), c_img_all=seen["c_img_all"].numpy(), occ_new=seen["occ_new"].numpy(),
             loss_pretrained=np.array(losses[True]), loss_joint=np.array(losses[False]),
             loss_plain_pretrained=np.array(plain[True]), loss_plain_joint=np.array(plain[False]), p_sample_plain=seen["p_sample_plain"].numpy(), num_sample=np.array(NS), seed=np.array(77),
             cube_v=m["cube"]["v"], cube_f=m["cube"]["f"], tet_v=m["tet"]["v"], tet_f=m["tet"]["f"])


if __name__ == "__main__":
    main()
