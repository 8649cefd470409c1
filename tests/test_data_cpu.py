"""Dataset layer (SURVEY.md section 8f "next" row 4) against the reference's own loader: the same synthetic
dataset files, the same numpy seed before every sample -> the same sample, bit for bit
(tests/golden/g9_data.npz, written by tests/golden/make_data_goldens.py from the real reference)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from synth_dataset import make_cfg, make_synthetic_dataset

G = dict(np.load(os.path.join(GOLDEN, "g9_data.npz"), allow_pickle=True))
VARIANTS = {"f32": (False, False, 128), "f16packed": (True, True, 128), "balanced": (False, False, [40, 24])}


@pytest.mark.parametrize("variant", list(VARIANTS))
def test_samples_equal_reference_loader(variant, tmp_path):
    from vtaco_amd.config import get_dataset
    half, pack, sub = VARIANTS[variant]
    make_synthetic_dataset(str(tmp_path), seed=3, half_points=half, packbits=pack)
    cfg = make_cfg(str(tmp_path), points_subsample=sub, unpackbits=pack)
    cfg["data"]["classes"] = ["ycb", "akb"]
    checked = 0
    for mode in ("train", "val"):
        ds = get_dataset(mode, cfg, return_idx=True)
        assert len(ds) == int(G[f"{variant}.{mode}.len"])
        for i in range(len(ds) if variant == "f32" else 1):
            np.random.seed(100 + i)
            sample = ds[i]
            want = {k[len(f"{variant}.{mode}.{i}."):]: v for k, v in G.items() if k.startswith(f"{variant}.{mode}.{i}.")}
            assert set(sample) == set(want)
            for k, v in want.items():
                got = np.asarray(sample[k])
                assert got.dtype == v.dtype and got.shape == v.shape, (mode, i, k, got.dtype, v.dtype)
                assert np.array_equal(got, v), (mode, i, k)
                checked += 1
    assert checked > 20


def test_batches_collate_and_feed_the_model_contract(tmp_path):
    """DataLoader + collate_remove_none give the tensors the training step reads: inputs [B,T,3],
    points [B,N,3], points.occ [B,N]."""
    import torch
    from vtaco_amd import data
    from vtaco_amd.config import get_dataset
    make_synthetic_dataset(str(tmp_path), seed=1)
    cfg = make_cfg(str(tmp_path))
    ds = get_dataset("train", cfg)
    loader = torch.utils.data.DataLoader(ds, batch_size=3, shuffle=False, collate_fn=data.collate_remove_none,
                                         worker_init_fn=data.worker_init_fn)
    batch = next(iter(loader))
    assert batch["inputs"].shape == (3, 150, 3) and batch["inputs"].dtype == torch.float32
    assert batch["points"].shape == (3, 128, 3) and batch["points.occ"].shape == (3, 128)
    assert batch["inputs.img"].shape == (3, 5, 3, 4, 3) and float(batch["inputs.img"].max()) <= 1 / 255 + 1e-9
    assert batch["points.cam_rot"].abs().max() <= np.pi + 1e-6             # degrees on disk, radians in the sample


def test_missing_file_is_skipped_or_raised(tmp_path):
    from vtaco_amd import data
    from vtaco_amd.config import get_dataset
    make_synthetic_dataset(str(tmp_path), seed=2)
    os.remove(os.path.join(str(tmp_path), "ycb", "obj_a_0001", "points.npz"))
    cfg = make_cfg(str(tmp_path))
    ds = get_dataset("train", cfg)
    bad = [i for i, m in enumerate(ds.models) if m["model"] == "obj_a_0001"][0]
    assert ds[bad] is None                                                  # no_except: the collate function drops it
    assert len(data.collate_remove_none([ds[bad], ds[(bad + 1) % len(ds)]])["inputs"]) == 1
    ds.no_except = False
    with pytest.raises(Exception):
        ds[bad]
    assert not ds.test_model_complete("ycb", "obj_a_0001") and ds.test_model_complete("ycb", "obj_b_0002")


def test_compute_iou_matches_reference_semantics():
    from vtaco_amd.eval import chamfer_distance_naive, compute_iou
    import torch
    occ_hat = np.array([[0.9, 0.2, 0.7, 0.1], [0.1, 0.1, 0.9, 0.9]])
    occ = np.array([[1, 0, 0, 0], [0, 0, 1, 1]], dtype=np.float32)
    # the threshold argument is ignored: both sides are cut at mean(occ) = 0.375
    assert np.allclose(compute_iou(occ_hat, occ, 0.99), [0.5, 1.0])
    a = torch.rand(2, 64, 3, generator=torch.Generator().manual_seed(0))
    assert torch.allclose(chamfer_distance_naive(a, a), torch.zeros(2))
    b = a + 0.1
    d = chamfer_distance_naive(a, b)
    assert (d > 0).all() and torch.allclose(d, chamfer_distance_naive(b, a))


def test_mesh_dict_reads_off_and_obj(tmp_path):
    """vtaco_amd.data.load_mesh_dict: the vf_dict of train.py:161-174 (.off preferred, .obj fallback, polygons triangulated)."""
    import numpy as np
    from vtaco_amd.data import load_mesh_dict
    (tmp_path / "cube.off").write_text("OFF\n8 6 0\n" + "\n".join(f"{x} {y} {z}" for z in (0, 1) for y in (0, 1) for x in (0, 1)) +
                                       "\n4 0 2 3 1\n4 4 5 7 6\n4 0 1 5 4\n4 2 6 7 3\n4 0 4 6 2\n4 1 3 7 5\n")
    (tmp_path / "tet.obj").write_text("# a tetrahedron\nv 0 0 0\nv 1 0 0\nv 0 1 0\nv 0 0 1\nf 1/1 3/2 2/3\nf 1 2 4\nf 1 4 3\nf -3 -2 -1\n")
    d = load_mesh_dict(str(tmp_path), ["cube", "tet", "cube"])
    assert set(d) == {"cube", "tet"}
    assert d["cube"]["v"].dtype == np.float32 and d["cube"]["v"].shape == (8, 3) and d["cube"]["f"].shape == (12, 3)
    assert d["tet"]["f"].tolist() == [[0, 2, 1], [0, 1, 3], [0, 3, 2], [1, 2, 3]]
    import pytest
    with pytest.raises(FileNotFoundError):
        load_mesh_dict(str(tmp_path), ["sphere"])
