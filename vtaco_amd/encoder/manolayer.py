"""MANO hand layer (drop-in for reference src/encoder/manolayer.py:14-364, the manopth layer).

Built for the configuration the shipped configs use (configs/VTacO/VTacO_YCB.yaml:46-56: axis-angle
root and joint rotations, ``use_pca: False``, ``flat_hand_mean: False``, right hand); the PCA pose
space is kept because it is one matmul in front, the rotmat/quat input modes are refused loudly.

``vt_mano_fwd`` -- one HIP workgroup per hand -- runs the layer; under autograd its backward is
``vt_mano_bwd`` (d pose from d verts / d joints: the plumbing between ``loss_pc`` and ``fc_mano``,
training.py:493-494).  ``forward_torch`` is the same arithmetic as differentiable host PyTorch ops: the left
hand (the kernels carry the right hand's tip table) and the ``VTACO_MANO_BACKWARD=host`` A/B knob use it.

The model file is read WITHOUT chumpy: MANO_RIGHT.pkl holds one chumpy object (``shapedirs``, a
``chumpy.reordering.Select`` over a ``Ch``), resolved here from its pickled state.  The asset itself is
licensed (mano.is.tue.mpg.de) and is not part of this repository: pass ``mano_root``.
"""
from __future__ import annotations

import os
import pickle

import numpy as np
import torch
from torch import nn

from .. import ops
from .._lib import VtError

PARENTS = (-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11, 0, 13, 14)
TIPS = {"right": (745, 317, 444, 556, 673), "left": (745, 317, 445, 556, 673)}      # manolayer.py:327-330
JOINT_ORDER = (0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20)


class _ChumpyState:
    """Stand-in the unpickler builds for any ``chumpy.*`` class: keeps the pickled state."""

    def __init__(self, *args, **kwargs):
        self.state = {}

    def __setstate__(self, state):
        self.state = state if isinstance(state, dict) else {"value": state}

    def resolve(self):
        st = self.state
        if "x" in st:                                             # chumpy.ch.Ch: the array itself
            return np.asarray(st["x"], dtype=np.float64)
        if "a" in st and "idxs" in st:                            # chumpy.reordering.Select: a.ravel()[idxs]
            base = st["a"].resolve() if isinstance(st["a"], _ChumpyState) else np.asarray(st["a"])
            out = base.ravel()[np.asarray(st["idxs"])]
            shape = st.get("preferred_shape")
            return out.reshape(shape) if shape is not None else out
        raise VtError(f"load_mano_pkl: cannot resolve a chumpy object with state keys {sorted(st)}")


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module.split(".")[0] == "chumpy":
            return type(name, (_ChumpyState,), {})
        return super().find_class(module, name)


def load_mano_pkl(path):
    """MANO_{RIGHT,LEFT}.pkl -> dict of float64 numpy arrays (+ int faces / kintree), no chumpy needed."""
    if not os.path.exists(path):
        raise VtError(f"load_mano_pkl: {path} not found (the MANO asset is licensed and not shipped: "
                      "download it from mano.is.tue.mpg.de and point mano_root at its folder)")
    with open(path, "rb") as fh:
        dd = _Unpickler(fh, encoding="latin1").load()
    out = {}
    for key in ("v_template", "shapedirs", "posedirs", "weights", "hands_components", "hands_mean", "J_regressor",
                "f", "kintree_table"):
        if key not in dd:
            raise VtError(f"load_mano_pkl: {path} has no '{key}'")
        v = dd[key]
        if isinstance(v, _ChumpyState):
            v = v.resolve()
        elif hasattr(v, "toarray"):                               # scipy sparse J_regressor
            v = v.toarray()
        out[key] = np.asarray(v)
    out["betas"] = np.asarray(dd["betas"].resolve() if isinstance(dd.get("betas"), _ChumpyState)
                              else dd.get("betas", np.zeros(out["shapedirs"].shape[-1])), dtype=np.float64)
    parents = [int(p) for p in out["kintree_table"][0][1:]]
    if tuple(parents) != PARENTS[1:]:
        raise VtError(f"load_mano_pkl: unexpected kinematic tree {parents} (the layer is built for MANO's 16-joint hand)")
    return out


def _rodrigues(axisang):
    """[N,3] axis-angle -> [N,3,3] through the re-normalised quaternion (manopth/rodrigues_layer.py:15-60)."""
    angle = torch.norm(axisang + 1e-8, p=2, dim=1, keepdim=True)
    half = 0.5 * angle
    quat = torch.cat([torch.cos(half), torch.sin(half) * (axisang / angle)], dim=1)
    quat = quat / quat.norm(p=2, dim=1, keepdim=True)
    w, x, y, z = quat.unbind(dim=1)
    rows = [w * w + x * x - y * y - z * z, 2 * x * y - 2 * w * z, 2 * w * y + 2 * x * z,
            2 * w * z + 2 * x * y, w * w - x * x + y * y - z * z, 2 * y * z - 2 * w * x,
            2 * x * z - 2 * w * y, 2 * w * x + 2 * y * z, w * w - x * x - y * y + z * z]
    return torch.stack(rows, dim=1).view(-1, 3, 3)


class _ManoFn(torch.autograd.Function):
    """vt_mano_fwd under autograd: the backward is vt_mano_bwd (d pose from d verts, d joints; the model blob is a constant)."""

    @staticmethod
    def forward(ctx, pose48, blob, center_idx):
        ctx.save_for_backward(pose48, blob)
        ctx.center_idx = center_idx
        return ops.mano_fwd(pose48, blob, center_idx)

    @staticmethod
    def backward(ctx, dverts, djoints):
        pose48, blob = ctx.saved_tensors
        return ops.mano_bwd(pose48, blob, ctx.center_idx, dverts, djoints), None, None


class ManoLayer(nn.Module):
    """Constructor arguments as the reference (manolayer.py:28-41)."""

    def __init__(self, center_idx=None, flat_hand_mean=True, ncomps=6, side="right", mano_root="mano/models",
                 use_pca=True, root_rot_mode="axisang", joint_rot_mode="axisang", robust_rot=False,
                 return_transf=False, return_full_pose=False):
        super().__init__()
        if root_rot_mode not in ("axisang", "rotmat", "quat"):
            raise KeyError(f"root_rot_mode not found. shoule be one of 'axisang' or 'rotmat' or 'quat'. got {root_rot_mode}")
        if use_pca and joint_rot_mode != "axisang":
            raise TypeError(f"if use_pca, joint_rot_mode must be 'axisang'. got {joint_rot_mode}")
        if root_rot_mode != "axisang" or joint_rot_mode != "axisang":
            raise VtError("ManoLayer: only root_rot_mode='axisang' with joint_rot_mode='axisang' is built "
                          "(what configs/VTacO/*.yaml use); rotmat / quat inputs are not")
        if return_transf:
            raise VtError("ManoLayer: return_transf=True is not built (no caller on the VTacO path reads the transforms)")
        if side not in ("right", "left"):
            raise VtError(f"ManoLayer: side must be 'right' or 'left', got {side!r}")
        self.center_idx, self.robust_rot, self.flat_hand_mean = center_idx, robust_rot, flat_hand_mean
        self.return_transf, self.return_full_pose = return_transf, return_full_pose
        self.side, self.use_pca, self.joint_rot_mode, self.root_rot_mode = side, use_pca, joint_rot_mode, root_rot_mode
        self.rot = 3
        self.ncomps = ncomps if use_pca else 45
        self.mano_path = os.path.join(mano_root, "MANO_RIGHT.pkl" if side == "right" else "MANO_LEFT.pkl")
        dd = load_mano_pkl(self.mano_path)
        f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
        # buffer names and shapes of the reference (manolayer.py:120-141): they are part of the checkpoint
        self.register_buffer("th_betas", f32(dd["betas"]).unsqueeze(0))
        self.register_buffer("th_shapedirs", f32(dd["shapedirs"]))
        self.register_buffer("th_posedirs", f32(dd["posedirs"]))
        self.register_buffer("th_v_template", f32(dd["v_template"]).unsqueeze(0))
        self.register_buffer("th_J_regressor", f32(dd["J_regressor"]))
        self.register_buffer("th_weights", f32(dd["weights"]))
        self.register_buffer("th_faces", torch.from_numpy(dd["f"].astype(np.int32)).long())
        mean = np.zeros(45) if flat_hand_mean else dd["hands_mean"]
        self.register_buffer("th_hands_mean", f32(mean).unsqueeze(0))
        self.register_buffer("th_selected_comps", f32(dd["hands_components"][:ncomps]))
        self.kintree_parents = [int(p) for p in dd["kintree_table"][0]]
        self._blob = None

    def _apply(self, fn, *args, **kwargs):
        self._blob = None                                             # the packed model follows the buffers
        return super()._apply(fn, *args, **kwargs)

    def _packed(self):
        if self._blob is None or self._blob.device != self.th_posedirs.device:
            self._blob = ops.mano_pack(self.th_v_template[0], self.th_shapedirs, self.th_betas[0], self.th_posedirs,
                                       self.th_J_regressor, self.th_weights, self.th_hands_mean[0], left=self.side == "left")
        return self._blob

    def _axis_angles(self, th_pose_coeffs):
        """[B, 3 + ncomps] -> [B,48] root axis-angle + 45 joint angles WITHOUT hands_mean (added downstream)."""
        if th_pose_coeffs.dim() != 2 or th_pose_coeffs.shape[1] < self.rot + self.ncomps:
            raise VtError(f"ManoLayer: pose coefficients must be [B,{self.rot + self.ncomps}], got {tuple(th_pose_coeffs.shape)}")
        hand = th_pose_coeffs[:, self.rot:self.rot + self.ncomps]
        if self.use_pca:
            hand = hand.mm(self.th_selected_comps)
        return torch.cat([th_pose_coeffs[:, :self.rot], hand], dim=1)

    def forward_torch(self, pose48):
        """Differentiable host-PyTorch form of vt_mano_fwd (same arithmetic, manolayer.py:186-347)."""
        B = pose48.shape[0]
        full = torch.cat([pose48[:, :3], self.th_hands_mean + pose48[:, 3:]], dim=1)
        rots = _rodrigues(full.reshape(-1, 3)).view(B, 16, 3, 3)
        pose_map = (rots[:, 1:] - torch.eye(3, device=pose48.device)).reshape(B, 135)
        v_shaped = torch.matmul(self.th_shapedirs, self.th_betas[0]) + self.th_v_template[0]
        J = torch.matmul(self.th_J_regressor, v_shaped)
        v_posed = v_shaped.unsqueeze(0) + torch.matmul(pose_map, self.th_posedirs.reshape(-1, 135).t()).view(B, -1, 3)
        par = torch.tensor(PARENTS[1:], device=J.device)
        rel = torch.cat([J[:1], J[1:] - J[par]], dim=0)                               # [16,3]
        Rg, tg = [rots[:, 0]], [rel[0].expand(B, 3)]
        for j in range(1, 16):
            p = PARENTS[j]
            Rg.append(torch.matmul(Rg[p], rots[:, j]))
            tg.append(torch.matmul(Rg[p], rel[j].view(1, 3, 1)).squeeze(-1) + tg[p])
        Rg, tg = torch.stack(Rg, dim=1), torch.stack(tg, dim=1)                       # [B,16,3,3], [B,16,3]
        ta = tg - torch.matmul(Rg, J.view(1, 16, 3, 1)).squeeze(-1)                   # rest pose removed
        Rv = torch.einsum("vj,bjrc->bvrc", self.th_weights, Rg)
        tv = torch.einsum("vj,bjr->bvr", self.th_weights, ta)
        verts = torch.matmul(Rv, v_posed.unsqueeze(-1)).squeeze(-1) + tv
        tips = verts[:, list(TIPS[self.side])]
        jtr = torch.cat([tg, tips], dim=1)[:, list(JOINT_ORDER)]
        if self.center_idx is not None:
            centre = jtr[:, self.center_idx].unsqueeze(1)
            verts, jtr = verts - centre, jtr - centre
        return verts, jtr

    def forward(self, th_pose_coeffs, th_betas=None, th_trans=None, root_palm=None, share_betas=None):
        if th_betas is not None or th_trans is not None or root_palm is not None or share_betas is not None:
            raise VtError("ManoLayer: per-call th_betas / th_trans / root_palm / share_betas are not built "
                          "(the VTacO path calls the layer with the pose only, pointnet.py:198, 206)")
        if not th_pose_coeffs.is_cuda:
            raise VtError(f"ManoLayer: inputs must live on a HIP device (got {th_pose_coeffs.device})")
        pose48 = self._axis_angles(th_pose_coeffs.float())
        if torch.is_grad_enabled() and pose48.requires_grad:
            if os.environ.get("VTACO_MANO_BACKWARD", "hip") == "host":
                verts, jtr = self.forward_torch(pose48)              # A/B knob: host-PyTorch autograd
            else:
                verts, jtr = _ManoFn.apply(pose48.contiguous(), self._packed(), self.center_idx)
        else:
            verts, jtr = ops.mano_fwd(pose48, self._packed(), self.center_idx)
        results = [verts, jtr]
        if self.return_full_pose:
            results.append(torch.cat([pose48[:, :3], self.th_hands_mean + pose48[:, 3:]], dim=1))
        return tuple(results)
