"""CPU: oracle/mc_lewiner.c against scikit-image 0.18.3's own outputs (g7_mc.npz).
Faces (vertex numbering included) must be bit-exact; vertex coordinates <= 1e-5."""
import numpy as np
import pytest

from conftest import GOLDEN
from oracle import mc

_Z = np.load(GOLDEN + "/g7_mc.npz")
Z = {k: _Z[k] for k in _Z.files}
CASES = sorted({k.rsplit(".", 1)[0] for k in Z} - {"cells"})


@pytest.mark.parametrize("name", CASES)
def test_matches_skimage(name):
    vol, level = Z[name + ".vol"], float(Z[name + ".level"])
    explicit = "@" in name
    verts, faces, lvl = mc.marching_cubes(vol, level if explicit else None)
    assert lvl == level
    assert faces.shape == Z[name + ".faces"].shape and verts.shape == Z[name + ".verts"].shape
    assert np.array_equal(faces, Z[name + ".faces"])
    assert np.abs(verts - Z[name + ".verts"]).max() <= 1e-5


def test_no_surface_raises():
    with pytest.raises(RuntimeError):
        mc.marching_cubes(np.zeros((4, 4, 4), np.float32), 1.0)


def test_single_cells_every_subcase_and_ties():
    """4000 one-cell volumes (random, tiny-valued, and integer-valued with exact ties):
    exercises every MC33 branch (face tests, interior tests, centre vertex)."""
    vols, nf, nv = Z["cells.vols"], Z["cells.nf"], Z["cells.nv"]
    seen = 0
    for i in range(len(vols)):
        try:
            v, f, _ = mc.marching_cubes(vols[i], 0.0)
        except RuntimeError:
            assert nf[i] == 0
            continue
        assert len(f) == nf[i] and len(v) == nv[i], i
        assert np.array_equal(f, Z["cells.faces"][i, :nf[i]]), i
        assert np.abs(v - Z["cells.verts"][i, :nv[i]]).max() <= 1e-5, i
        seen += 1
    assert seen > 3500
