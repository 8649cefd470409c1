"""Object meshes for the VTacO training step's occupancy labels: the dictionary the reference builds in train.py:161-174
(``vf_dict[obj_name] = {'v': float32 [V,3], 'f': int [F,3]}`` from ``<root>/<obj_name>.off`` or ``.obj``, read there with
``igl.read_triangle_mesh``).  Plain-text OFF and OBJ readers for triangle meshes (polygons are fan-triangulated)."""
from __future__ import annotations

import os

import numpy as np


def _fan(poly):
    return [(poly[0], poly[i], poly[i + 1]) for i in range(1, len(poly) - 1)]


def read_off(path):
    with open(path) as fh:
        tokens = fh.read().split()
    if not tokens or not tokens[0].upper().startswith("OFF"):
        raise ValueError(f"{path}: not an OFF file")
    head = tokens[0][3:]
    pos = 1
    if head:                                                      # 'OFF123 456 0' written without a separator
        tokens = [head] + tokens[1:]
        pos = 0
    nv, nf = int(tokens[pos]), int(tokens[pos + 1])
    pos += 3
    verts = np.array(tokens[pos:pos + 3 * nv], dtype=np.float64).reshape(nv, 3)
    pos += 3 * nv
    faces = []
    for _ in range(nf):
        k = int(tokens[pos])
        faces += _fan([int(t) for t in tokens[pos + 1:pos + 1 + k]])
        pos += 1 + k
    return verts, np.array(faces, dtype=np.int64).reshape(-1, 3)


def read_obj(path):
    verts, faces = [], []
    with open(path) as fh:
        for line in fh:
            parts = line.split()
            if not parts:
                continue
            if parts[0] == "v":
                verts.append([float(x) for x in parts[1:4]])
            elif parts[0] == "f":
                idx = [int(tok.split("/")[0]) for tok in parts[1:]]
                idx = [i - 1 if i > 0 else len(verts) + i for i in idx]      # 1-based, negative = relative
                faces += _fan(idx)
    return np.array(verts, dtype=np.float64).reshape(-1, 3), np.array(faces, dtype=np.int64).reshape(-1, 3)


def read_triangle_mesh(path):
    ext = os.path.splitext(path)[1].lower()
    if ext == ".off":
        return read_off(path)
    if ext == ".obj":
        return read_obj(path)
    raise ValueError(f"{path}: only .off and .obj meshes are read")


def load_mesh_dict(root, names):
    """{name: {'v': float32 [V,3], 'f': int64 [F,3]}} for every object name (``.off`` first, then ``.obj``: train.py:166-169)."""
    out = {}
    for name in names:
        if name in out:
            continue
        path = os.path.join(root, name + ".off")
        if not os.path.exists(path):
            path = os.path.join(root, name + ".obj")
        if not os.path.exists(path):
            raise FileNotFoundError(f"load_mesh_dict: neither {name}.off nor {name}.obj under {root}")
        v, f = read_triangle_mesh(path)
        out[name] = {"v": v.astype(np.float32), "f": f}
    return out
