"""A seeded, synthetic MANO-format hand model (tests, and the synthetic training workload of bench.py).

The real MANO_RIGHT.pkl is licensed and is never committed; the tests only need an asset with the same
keys, shapes and structure (778 vertices, 16 joints in three-per-finger chains, 135 pose blend shapes).
``make_asset`` is deterministic (numpy RandomState), so the golden generator and the tests rebuild the
same arrays and the fixture stores outputs only.  ``write_pkl`` stores it the way MANO ships: a pickled
dict with a scipy-sparse J_regressor (plain arrays otherwise).
"""
import os
import pickle

import numpy as np

PARENTS = [-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11, 0, 13, 14]


def make_asset(seed=0):
    rs = np.random.RandomState(seed)
    nv = 778
    v_template = (rs.rand(nv, 3) - 0.5) * np.array([0.18, 0.08, 0.04])
    shapedirs = rs.randn(nv, 3, 10) * 0.004
    posedirs = rs.randn(nv, 3, 135) * 0.002
    jr = np.zeros((16, nv))
    for j in range(16):                                        # each joint regressed from ~12 vertices
        sel = rs.choice(nv, 12, replace=False)
        w = rs.rand(12)
        jr[j, sel] = w / w.sum()
    weights = np.zeros((nv, 16))
    for v in range(nv):                                        # each vertex skinned to 1-4 joints
        k = rs.randint(1, 5)
        sel = rs.choice(16, k, replace=False)
        w = rs.rand(k) + 0.1
        weights[v, sel] = w / w.sum()
    return {
        "v_template": v_template, "shapedirs": shapedirs, "posedirs": posedirs,
        "J_regressor": jr, "weights": weights,
        "f": rs.randint(0, nv, size=(1538, 3)).astype(np.uint32),
        "hands_components": rs.randn(45, 45) * 0.3, "hands_mean": rs.randn(45) * 0.2,
        "hands_coeffs": rs.randn(8, 45),
        "kintree_table": np.array([[4294967295] + PARENTS[1:], list(range(16))], dtype=np.int64),
        "bs_type": "lrotmin", "bs_style": "lbs",
    }


def write_pkl(asset, root, side="right"):
    import scipy.sparse as sp
    dd = dict(asset)
    dd["J_regressor"] = sp.csc_matrix(asset["J_regressor"])
    os.makedirs(root, exist_ok=True)
    path = os.path.join(root, "MANO_RIGHT.pkl" if side == "right" else "MANO_LEFT.pkl")
    with open(path, "wb") as fh:
        pickle.dump(dd, fh, protocol=2)
    return path


def as_model(asset):
    """The f32 tensors ``oracle.mano_forward`` takes (betas = the model's zeros, manolayer.py:118-127)."""
    import torch
    t = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32))
    return {"v_template": t(asset["v_template"]), "shapedirs": t(asset["shapedirs"]),
            "posedirs": t(asset["posedirs"]), "J_regressor": t(asset["J_regressor"]),
            "weights": t(asset["weights"]), "hands_mean": t(asset["hands_mean"]),
            "betas": torch.zeros(10)}
