#!/usr/bin/env python3
"""Golden vectors for the tactile training-sample assembly of the VTacOH trainer (g11_trainer_img.npz) from the REAL
reference: ``Trainer.compute_loss_img`` (src/conv_onet/training.py:502-626) run on a stand-in model that returns seeded
tensors and records what the trainer hands to ``decode_img``.

Build container only.  src/conv_onet/training.py imports igl / trimesh (not installed; never reached on this path) and
reads a dataset file at import: stand-ins for the two modules, and np.loadtxt answers that one read with zeros.

    python tests/golden/make_trainer_goldens.py
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


def main():
    mg._install_stubs()
    for name in ("igl", "trimesh", "skimage", "skimage.measure"):
        sys.modules.setdefault(name, types.ModuleType(name))
    if "matplotlib" not in sys.modules:
        try:
            import matplotlib  # noqa: F401
        except Exception:
            mpl = types.ModuleType("matplotlib")
            sys.modules.update({"matplotlib": mpl, "matplotlib.pyplot": types.ModuleType("matplotlib.pyplot"),
                                "mpl_toolkits": types.ModuleType("mpl_toolkits"),
                                "mpl_toolkits.mplot3d": types.ModuleType("mpl_toolkits.mplot3d")})
            sys.modules["mpl_toolkits.mplot3d"].Axes3D = object
    loadtxt = np.loadtxt
    np.loadtxt = lambda *a, **k: np.zeros((320, 240))
    try:
        training = importlib.import_module("src.conv_onet.training")
    finally:
        np.loadtxt = loadtxt
    from src.common import R_from_PYR, norm_pc_1

    B, N, NS = 2, 3000, 1024
    g = torch.Generator().manual_seed(40)
    mano_gt = torch.randn(B, 51, generator=g) * 0.2
    wrist = torch.randn(B, 3, generator=g) * 0.5
    pc_ply = torch.randn(B, 500, 3, generator=g) * 0.2 + 0.05
    pc_hand = torch.randn(B, 778, 3, generator=g) * 0.05
    joints = torch.randn(B, 21, 3, generator=g) * 0.05
    hand = {"mano_param": torch.randn(B, 51, generator=g) * 0.2, "mano_verts": torch.randn(B, 778, 3, generator=g) * 0.05,
            "mano_joints": joints}
    c_img = torch.randn(B, 5, 32, generator=g)
    touch = torch.tensor([[True, True, False, True, True], [True, False, True, True, True]])
    # where the trainer will see the five fingertips (its own frame change): put query points around them
    tips = np.zeros((B, 5, 3), dtype=np.float32)
    for b in range(B):
        t = joints[b, [4, 8, 12, 16, 20]].numpy() - np.array([0.11, 0.005, 0], dtype=np.float32)
        t = np.linalg.inv(R_from_PYR(np.array([-np.pi / 2, np.pi / 2, 0]))) @ t.T
        t = np.linalg.inv(R_from_PYR(np.array(wrist[b].numpy()))) @ t
        tips[b] = norm_pc_1(t.T + mano_gt[b, :3].numpy(), pc_ply[b].numpy())
    p = (torch.rand(B, N, 3, generator=g) - 0.5) * 1.1
    near = [700, 60, 40, 0, 25]                       # finger 0 gets more than the 512-point cap
    for b in range(B):
        k = 0
        for f in range(5):
            d = torch.randn(near[f], 3, generator=g)
            d = d / d.norm(dim=1, keepdim=True) * torch.rand(near[f], 1, generator=g) * 0.06      # some just outside 0.05
            p[b, k:k + near[f]] = torch.from_numpy(tips[b, f]) + d
            k += near[f]
        p[b] = p[b, torch.randperm(N, generator=g)]
    occ = (torch.rand(B, N, generator=g) < 0.4).float()
    data = {"points": p, "points.occ": occ, "points.mano": mano_gt, "points.pc_hand": pc_hand, "points.wrist": wrist,
            "inputs": torch.zeros(B, 16, 3), "inputs.pc_ply": pc_ply, "inputs.img": torch.zeros(B, 5, 3, 8, 6),
            "inputs.touch_success": touch}
    seen = {}

    class FakeModel(object):
        def encode_inputs(self, inputs):
            return "c"

        def encode_hand_inputs(self, inputs):
            return hand

        def encode_img_inputs(self, imgs):
            return c_img

        def decode_img(self, p_sample, c, c_img_all, **kw):
            seen["p_sample"], seen["c_img_all"] = p_sample.detach().clone(), c_img_all.detach().clone()
            return types.SimpleNamespace(logits=p_sample.sum(-1) * 0.5 + c_img_all.sum(-1) * 0.1)

    l1 = training.F.l1_loss

    def spy_l1(a, b, *args, **kw):
        seen["occ_new"] = b.detach().clone()
        return l1(a, b, *args, **kw)

    trainer = training.Trainer(FakeModel(), None, device="cpu", num_sample=NS, with_img=True)
    np.random.seed(123)
    training.F.l1_loss = spy_l1
    # the trainer builds `torch.zeros(.., requires_grad=True).to(device)` and writes into it: on a GPU `.to` copies (a non-leaf,
    # writable); on the CPU it returns the leaf itself and autograd refuses the write.  Give the CPU run the GPU's behaviour.
    to = torch.Tensor.to
    torch.Tensor.to = lambda self, *a, **k: (to(self, *a, **k).clone() if self.is_leaf and self.requires_grad else to(self, *a, **k))
    try:
        loss, loss_mano, loss_pc = trainer.compute_loss_img(data)
    finally:
        training.F.l1_loss = l1
        torch.Tensor.to = to
    filled = (seen["c_img_all"].abs().sum(-1) > 0).sum(1)
    print("tactile rows per scene:", filled.tolist(), "loss %.6f mano %.6f pc %.6f" % (float(loss), float(loss_mano), float(loss_pc)))
    mg._save("g11_trainer_img.npz", p=p.numpy(), occ=occ.numpy(), mano=mano_gt.numpy(), wrist=wrist.numpy(), pc_ply=pc_ply.numpy(),
             pc_hand=pc_hand.numpy(), touch=touch.numpy(), c_img=c_img.numpy(), mano_param=hand["mano_param"].numpy(),
             mano_verts=hand["mano_verts"].numpy(), mano_joints=joints.numpy(), tips=tips,
             p_sample=seen["p_sample"].numpy(), c_img_all=seen["c_img_all"].numpy().astype(np.float32), occ_new=seen["occ_new"].numpy(),
             loss=np.array([float(loss), float(loss_mano), float(loss_pc)]), num_sample=np.array(NS), seed=np.array(123))


if __name__ == "__main__":
    main()
