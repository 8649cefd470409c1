#!/opt/conda/bin/python3.9
"""Marching-cubes golden vectors from scikit-image 0.18.3 itself (the third-party
dependency the reference calls at src/conv_onet/generation.py:270).  Runs only in
the build container:  /opt/conda/bin/python3.9 tests/golden/make_mc_goldens.py
Writes tests/golden/g7_mc.npz: for each case the float32 volume, the level used,
and skimage's verts (float32, array-axis order) and faces (int32)."""
import os

import numpy as np
from skimage import measure

OUT = os.path.dirname(os.path.abspath(__file__))
rng = np.random.RandomState(0)


def sphere(n, r=0.35):
    g = np.linspace(-0.5, 0.5, n, dtype=np.float32)
    x, y, z = np.meshgrid(g, g, g, indexing="ij")
    return (np.sqrt(x * x + y * y + z * z) - r).astype(np.float32)


def torus(n):
    g = np.linspace(-1, 1, n, dtype=np.float32)
    x, y, z = np.meshgrid(g, g, g, indexing="ij")
    return ((np.sqrt(x * x + y * y) - 0.6) ** 2 + z * z - 0.09).astype(np.float32)


vols = {
    "sphere16": sphere(16),
    "torus24": torus(24),
    "noise12": rng.randn(12, 12, 12).astype(np.float32),
    "noise20": rng.randn(20, 20, 20).astype(np.float32),
    "noise_rect": rng.randn(9, 14, 7).astype(np.float32),
    "smoothnoise24": None,
    "plateau10": np.round(rng.randn(10, 10, 10) * 1.5).astype(np.float32),   # many exact ties with the level
    "tiny2": rng.randn(2, 2, 2).astype(np.float32),
    "checker8": ((np.indices((8, 8, 8)).sum(0) % 2) * 2 - 1).astype(np.float32) * (1 + rng.rand(8, 8, 8)).astype(np.float32),
}
n = 24
k = rng.randn(n, n, n).astype(np.float32)
for _ in range(2):     # box-blurred noise: ambiguous cells with smooth values
    k = (k + np.roll(k, 1, 0) + np.roll(k, 1, 1) + np.roll(k, 1, 2)) / 4
vols["smoothnoise24"] = k.astype(np.float32)
p = os.path.join(OUT, "_mc_vol_logits32.npy")
if os.path.exists(p):
    vols["logits32"] = np.load(p).astype(np.float32)

out = {}
for name, vol in vols.items():
    lvl = None
    verts, faces, _, _ = measure.marching_cubes(vol, gradient_direction="ascent")
    level = 0.5 * (vol.min() + vol.max())
    out[name + ".vol"] = vol
    out[name + ".level"] = np.float64(level)
    out[name + ".verts"] = verts.astype(np.float32)
    out[name + ".faces"] = faces.astype(np.int32)
    print(name, vol.shape, "level", float(level), "V", len(verts), "F", len(faces))
    if name in ("noise20", "sphere16"):         # an explicit level as well
        verts, faces, _, _ = measure.marching_cubes(vol, 0.25, gradient_direction="ascent")
        out[name + "@0.25.vol"] = vol
        out[name + "@0.25.level"] = np.float64(0.25)
        out[name + "@0.25.verts"] = verts.astype(np.float32)
        out[name + "@0.25.faces"] = faces.astype(np.int32)
# single-cell volumes: every Lewiner case/sub-case incl. exact ties with the level
K = 4000
cells = rng.randn(K, 2, 2, 2).astype(np.float32)
cells[1::4] *= rng.rand(K // 4, 1, 1, 1).astype(np.float32) * 0.01
cells[2::4] = np.round(cells[2::4] * 1.2)
c_nf = np.zeros(K, np.int8)
c_nv = np.zeros(K, np.int8)
c_faces = -np.ones((K, 12, 3), np.int8)
c_verts = np.zeros((K, 16, 3), np.float32)
for i in range(K):
    try:
        v, f, _, _ = measure.marching_cubes(cells[i], 0.0, gradient_direction="ascent")
    except (RuntimeError, ValueError):
        continue
    c_nf[i], c_nv[i] = len(f), len(v)
    c_faces[i, :len(f)] = f
    c_verts[i, :len(v)] = v
out.update({"cells.vols": cells, "cells.nf": c_nf, "cells.nv": c_nv, "cells.faces": c_faces, "cells.verts": c_verts})
np.savez_compressed(os.path.join(OUT, "g7_mc.npz"), **out)
print("g7_mc.npz", os.path.getsize(os.path.join(OUT, "g7_mc.npz")) // 1024, "KiB")
