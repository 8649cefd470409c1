"""Evaluation metrics of the reference (src/common.py:11-91), host-side."""
from __future__ import annotations

import numpy as np
import torch


def compute_iou(occ1, occ2, threshold=None):
    """IoU of two occupancy sets (common.py:11-43).  As in the reference the ``threshold`` argument is
    IGNORED: both sets are binarised at mean(occ2)."""
    occ1, occ2 = np.asarray(occ1), np.asarray(occ2)
    if occ1.ndim >= 2:
        occ1 = occ1.reshape(occ1.shape[0], -1)
    if occ2.ndim >= 2:
        occ2 = occ2.reshape(occ2.shape[0], -1)
    thr = np.mean(occ2)
    a, b = occ1 >= thr, occ2 >= thr
    union = (a | b).astype(np.float32).sum(axis=-1)
    inter = (a & b).astype(np.float32).sum(axis=-1)
    return inter / union


def chamfer_distance_naive(points1, points2):
    """Squared-distance Chamfer of two [B,T,3] tensors (common.py:62-83); runs on whatever device they live on."""
    if points2.size(1) < 2048:
        points1 = points1[:, :points2.size(1), :]
    assert points1.size() == points2.size()
    d = (points1.unsqueeze(2) - points2.unsqueeze(1)).pow(2).sum(-1)
    return d.min(dim=1)[0].mean(dim=1) + d.min(dim=2)[0].mean(dim=1)


def earth_mover_distance(points1, points2):
    """EMD by optimal assignment (common.py:45-51)."""
    from scipy.optimize import linear_sum_assignment
    from scipy.spatial import distance
    d = distance.cdist(points1, points2)
    rows, cols = linear_sum_assignment(d)
    return d[rows, cols].sum() / len(d)
