#!/usr/bin/env python3
"""Golden samples of the dataset layer (tests/golden/g9_data.npz) from the REAL reference loader.

Build container only (needs /root/reference).  Feeds the synthetic dataset of tests/synth_dataset.py to the
reference's own ``src.data`` classes, assembled exactly as ``src/config.py:get_dataset`` +
``conv_onet/config.py:get_data_fields`` assemble them (field order: points, points_iou, inputs, idx), with
``np.random.seed`` set before every sample, and stores what ``Shapes3dDataset.__getitem__`` returns.
No reference code is stored."""
import os
import sys
import tempfile
import types

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from synth_dataset import make_cfg, make_synthetic_dataset  # noqa: E402


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m


for name in ("trimesh", "pykdtree", "pybullet", "torch_scatter"):
    _stub(name)
_stub("pykdtree.kdtree", KDTree=object)
sys.modules["pybullet"].computeProjectionMatrixFOV = lambda *a: [0.0] * 16
sys.path.insert(0, "/root/reference")
from src import data as rdata  # noqa: E402


class Compose:
    def __init__(self, ts):
        self.ts = ts

    def __call__(self, x):
        for t in self.ts:
            x = t(x)
        return x


def ref_dataset(mode, cfg):
    d = cfg["data"]
    fields = {"points": rdata.PointsField(d["points_file"], rdata.SubsamplePoints(d["points_subsample"]),
                                          unpackbits=d["points_unpackbits"], multi_files=d["multi_files"])}
    if mode in ("val", "test"):
        fields["points_iou"] = rdata.PointsField(d["points_iou_file"], unpackbits=d["points_unpackbits"],
                                                 multi_files=d["multi_files"])
    fields["inputs"] = rdata.PointCloudField(d["pointcloud_file"], Compose([rdata.SubsamplePointcloud(d["pointcloud_n"]),
                                                                            rdata.PointcloudNoise(d["pointcloud_noise"])]),
                                             multi_files=d["multi_files"])
    fields["idx"] = rdata.IndexField()
    split = {"train": d["train_split"], "val": d["val_split"]}[mode]
    return rdata.Shapes3dDataset(d["path"], fields, split=split, categories=["ycb", "akb"], cfg=cfg)


out = {}
for variant, (half, pack, sub) in {"f32": (False, False, 128), "f16packed": (True, True, 128), "balanced": (False, False, [40, 24])}.items():
    with tempfile.TemporaryDirectory() as root:
        make_synthetic_dataset(root, seed=3, half_points=half, packbits=pack)
        cfg = make_cfg(root, points_subsample=sub, unpackbits=pack)
        for mode in ("train", "val"):
            ds = ref_dataset(mode, cfg)
            out[f"{variant}.{mode}.len"] = np.array(len(ds))
            for i in range(len(ds) if variant == "f32" else 1):
                np.random.seed(100 + i)
                sample = ds[i]
                for k, v in sample.items():
                    out[f"{variant}.{mode}.{i}.{k}"] = np.asarray(v)
np.savez_compressed(os.path.join(HERE, "g9_data.npz"), **out)
print("wrote g9_data.npz:", len(out), "arrays,", os.path.getsize(os.path.join(HERE, "g9_data.npz")) // 1024, "KiB")
