"""A tiny synthetic dataset in the reference's on-disk layout (points.npz / pointcloud.npz per model,
<split>.lst per category, metadata.yaml), generated from a seed.  Used by tests/test_data_cpu.py and
by tests/golden/make_data_goldens.py (which feeds the same files to the reference's own loader)."""
import os

import numpy as np
import yaml

CATEGORIES = {"ycb": ["obj_a_0001", "obj_b_0002", "obj_c_0003"], "akb": ["cup_x_0001", "cup_y_0002"]}
SPLITS = {"ycb": {"train": ["obj_a_0001", "obj_b_0002"], "val": ["obj_c_0003"]},
          "akb": {"train": ["cup_x_0001"], "val": ["cup_y_0002"]}}


def make_synthetic_dataset(root, seed=0, half_points=False, packbits=False):
    rng = np.random.RandomState(seed)
    with open(os.path.join(root, "metadata.yaml"), "w") as fh:
        yaml.safe_dump({c: {"id": c, "name": c.upper()} for c in CATEGORIES}, fh)
    for cat, models in CATEGORIES.items():
        os.makedirs(os.path.join(root, cat), exist_ok=True)
        for split, names in SPLITS[cat].items():
            with open(os.path.join(root, cat, split + ".lst"), "w") as fh:
                fh.write("\n".join(names) + "\n")
        for m in models:
            d = os.path.join(root, cat, m)
            os.makedirs(d, exist_ok=True)
            P, T, M = 256, 200, 2100
            pts = (rng.rand(P, 3).astype(np.float32) - 0.5) * 1.1
            occ = (np.linalg.norm(pts, axis=1) < 0.3)
            np.savez(os.path.join(d, "points.npz"),
                     points=pts.astype(np.float16) if half_points else pts,
                     occupancies=np.packbits(occ) if packbits else occ.astype(np.uint8),
                     points_obj=rng.randn(M, 3).astype(np.float32) * 0.1,
                     contact=(rng.rand(P) < 0.1).astype(np.uint8),
                     pc_hand=rng.randn(778, 3).astype(np.float32) * 0.05,
                     mano=rng.randn(51).astype(np.float64),
                     wrist_rot=rng.randn(3).astype(np.float64),
                     cam_pos=rng.randn(5, 3).astype(np.float64),
                     cam_rot=(rng.rand(5, 3) * 360 - 180).astype(np.float64))
            surf = rng.randn(T, 3).astype(np.float32)
            surf = 0.3 * surf / np.linalg.norm(surf, axis=1, keepdims=True)
            np.savez(os.path.join(d, "pointcloud.npz"),
                     points=surf, normals=(surf / 0.3).astype(np.float32), pc_ply=rng.randn(100, 3).astype(np.float32),
                     img=rng.randint(0, 256, size=(5, 3, 4, 3)).astype(np.uint8),
                     depth=rng.rand(5, 4, 3).astype(np.float32), touch_success=np.array([1, 0, 1, 1, 0], dtype=np.uint8))


def make_cfg(root, points_subsample=128, unpackbits=False):
    return {"method": "conv_onet",
            "data": {"dataset": "Shapes3D", "path": root, "classes": None, "input_type": "pointcloud",
                     "train_split": "train", "val_split": "val", "test_split": "val", "dim": 3,
                     "points_file": "points.npz", "points_iou_file": "points.npz", "multi_files": None,
                     "points_subsample": points_subsample, "points_unpackbits": unpackbits, "voxels_file": None,
                     "pointcloud_file": "pointcloud.npz", "pointcloud_n": 150, "pointcloud_noise": 0.005, "padding": 0.1}}
