"""Host-side coordinate helpers with the reference's names (src/common.py).  The HIP
kernels re-implement this arithmetic in-kernel; these torch versions exist for callers
that want the explicit point tensors (e.g. ``eval_points`` on arbitrary points)."""
from __future__ import annotations

import torch


def make_3d_grid(bb_min, bb_max, shape):
    """Lattice with axis 0 slowest, axis 2 fastest (src/common.py:178-197)."""
    axes = [torch.linspace(bb_min[k], bb_max[k], shape[k]) for k in range(3)]
    mesh = torch.meshgrid(*axes, indexing="ij")
    return torch.stack([m.reshape(-1) for m in mesh], dim=1)


def normalize_3d_coordinate(p, padding=0.1):
    """src/common.py:293-309 (functional; the reference mutates a clone)."""
    q = p / (1 + padding + 10e-4) + 0.5
    q = torch.where(q >= 1, torch.full_like(q, 1 - 10e-4), q)
    return torch.where(q < 0, torch.zeros_like(q), q)


def coordinate2index(x, reso, coord_type='3d'):
    """src/common.py:333-348, returns [B,1,T] int64."""
    xi = (x * reso).long()
    if coord_type == '2d':
        index = xi[:, :, 0] + reso * xi[:, :, 1]
    else:
        index = xi[:, :, 0] + reso * (xi[:, :, 1] + reso * xi[:, :, 2])
    return index[:, None, :]


def rot_from_pyr(roll, pitch, yaw):
    """The reference's wrist-frame convention (R_from_PYR, src/common.py:591-604): a z rotation by ``roll`` applied first,
    then the TRANSPOSED y rotation by ``yaw``, then the transposed x rotation by ``pitch``.  3x3 numpy, host side."""
    import numpy as np
    cr, sr, cp, sp, cy, sy = np.cos(roll), np.sin(roll), np.cos(pitch), np.sin(pitch), np.cos(yaw), np.sin(yaw)
    about_z = np.array([[cr, -sr, 0], [sr, cr, 0], [0, 0, 1]])
    about_x_t = np.array([[1, 0, 0], [0, cp, sp], [0, -sp, cp]])
    about_y_t = np.array([[cy, 0, -sy], [0, 1, 0], [sy, 0, cy]])
    return about_x_t @ about_y_t @ about_z


def fingertips_in_object_frame(mano_joints, wrist_pos, wrist_euler, pc_ply):
    """The five fingertip joints (MANO joints 4, 8, 12, 16, 20) of every scene in the object's normalised frame
    (training.py:543-556, generation.py:178-184): out of the MANO frame (fixed offset and rotation), out of the wrist
    rotation, plus the wrist position, normalised like the object cloud (norm_pc_1, common.py:606-612).  5 x 3 numbers
    per scene, host-side numpy in the reference's operation order.  [B,21,3], [B,3], [B,3], [B,M,3] -> float64 [B,5,3]."""
    import numpy as np
    joints = np.asarray(mano_joints, dtype=np.float32)[:, [4, 8, 12, 16, 20]]
    fixed = np.linalg.inv(rot_from_pyr(-np.pi / 2, np.pi / 2, 0.0))
    tips = np.empty(joints.shape, dtype=np.float64)
    for b in range(joints.shape[0]):
        t = joints[b] - np.array([0.11, 0.005, 0], dtype=np.float32)
        t = np.linalg.inv(rot_from_pyr(*np.asarray(wrist_euler[b]))) @ (fixed @ t.T)
        t = t.T + np.asarray(wrist_pos[b])
        cloud = np.asarray(pc_ply[b])
        centroid = np.mean(cloud, axis=0)
        m = np.max(np.sqrt(np.sum((cloud - centroid) ** 2, axis=1)))
        tips[b] = (t - centroid) / (2 * m)
    return tips
