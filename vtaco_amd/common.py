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
