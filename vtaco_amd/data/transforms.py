"""Sample transforms (reference src/data/transforms.py)."""
from __future__ import annotations

import numpy as np


class PointcloudNoise:
    """points += stddev * N(0,1)  (transforms.py:5-28)."""

    def __init__(self, stddev):
        self.stddev = stddev

    def __call__(self, data):
        out = data.copy()
        pts = data[None]
        out[None] = pts + (self.stddev * np.random.randn(*pts.shape)).astype(np.float32)
        return out


class SubsamplePointcloud:
    """N surface points (and their normals) drawn with replacement (transforms.py:30-57)."""

    def __init__(self, N):
        self.N = N

    def __call__(self, data):
        out = data.copy()
        idx = np.random.randint(data[None].shape[0], size=self.N)
        out[None] = data[None][idx, :]
        out['normals'] = data['normals'][idx, :]
        return out


class SubsamplePoints:
    """N query points with their occupancy and contact labels, or (N_out, N_in) balanced between
    empty and occupied points (transforms.py:60-112)."""

    def __init__(self, N):
        self.N = N

    def __call__(self, data):
        pts, occ = data[None], data['occ']
        out = data.copy()
        if isinstance(self.N, int):
            idx = np.random.randint(pts.shape[0], size=self.N)
            out.update({None: pts[idx, :], 'occ': occ[idx], 'contact': data['contact'][idx]})
            return out
        n_out, n_in = self.N
        inside = occ >= 0.5
        p_out, p_in = pts[~inside], pts[inside]
        i_out = np.random.randint(p_out.shape[0], size=n_out)
        i_in = np.random.randint(p_in.shape[0], size=n_in)
        out.update({
            None: np.concatenate([p_out[i_out, :], p_in[i_in, :]], axis=0),
            'occ': np.concatenate([np.zeros(n_out, dtype=np.float32), np.ones(n_in, dtype=np.float32)], axis=0),
            'volume': (inside.sum() / len(inside)).astype(np.float32),
        })
        return out
