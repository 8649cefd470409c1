"""Fields of the VTacO datasets (reference src/data/fields.py).  Schema:

``points.npz``      points [P,3] f16/f32, occupancies [P] (or bit-packed with ``unpackbits``),
                    points_obj [M,3], contact, pc_hand, mano, wrist_rot, cam_pos, cam_rot (DEGREES)
``pointcloud.npz``  points [T,3], normals [T,3], pc_ply, img [5,3,H,W] 0..255, depth, touch_success
"""
from __future__ import annotations

import os

import numpy as np

from .core import Field


def _pick_file(model_path, file_name, multi_files):
    if multi_files is None:
        return os.path.join(model_path, file_name)
    num = np.random.randint(multi_files)
    return os.path.join(model_path, file_name, '%s_%02d.npz' % (file_name, num))


class IndexField(Field):
    """The sample's index (fields.py:12-30)."""

    def load(self, model_path, idx, category):
        return idx

    def check_complete(self, files):
        return True


class PointsField(Field):
    """Query points with occupancies and the hand / camera annotations (fields.py:99-177)."""

    def __init__(self, file_name, transform=None, unpackbits=False, multi_files=None):
        self.file_name, self.transform = file_name, transform
        self.unpackbits, self.multi_files = unpackbits, multi_files

    def load(self, model_path, idx, category):
        name = model_path.split("/")[-1][:-5]                     # the reference strips a 5-character suffix
        z = np.load(_pick_file(model_path, self.file_name, self.multi_files), allow_pickle=True)
        points = z['points']
        if points.dtype == np.float16:                            # break the symmetry of half-precision storage
            points = points.astype(np.float32)
            points += 1e-4 * np.random.randn(*points.shape)
        occ = z['occupancies']
        if self.unpackbits:
            occ = np.unpackbits(occ)[:points.shape[0]]
        occ = occ.astype(np.float32)
        points_obj = z['points_obj'].astype(np.float32)
        np.random.shuffle(points_obj)
        out = {
            None: points,
            'name': name,
            'occ': occ,
            'points_obj': points_obj[:2048],
            'contact': z['contact'].astype(np.float32),
            'pc_hand': z['pc_hand'].astype(np.float32),
            'mano': z['mano'].astype(np.float32),
            'wrist': z['wrist_rot'].astype(np.float32),
            'cam_pos': z['cam_pos'].astype(np.float32),
            'cam_rot': z['cam_rot'].astype(np.float32) / 180 * np.pi,
        }
        return self.transform(out) if self.transform is not None else out

    def check_complete(self, files):
        return self.file_name in files


class PointCloudField(Field):
    """Surface samples plus the five tactile images (fields.py:296-352).  The images get N(0, 7) pixel noise
    and are divided by 255 TWICE, as the reference does (:335-337) -- the tactile encoder was trained on that."""

    def __init__(self, file_name, transform=None, multi_files=None):
        self.file_name, self.transform, self.multi_files = file_name, transform, multi_files

    def load(self, model_path, idx, category):
        z = np.load(_pick_file(model_path, self.file_name, self.multi_files), allow_pickle=True)
        images = z['img']
        noise = np.random.normal(0, 7, images.shape)
        images = np.clip(images + noise, 0, 255) / 255
        out = {
            None: z['points'].astype(np.float32),
            'normals': z['normals'].astype(np.float32),
            'pc_ply': z['pc_ply'].astype(np.float32),
            'touch_success': z['touch_success'],
            'img': images / 255,
            'depth': z['depth'].astype(np.float32),
        }
        return self.transform(out) if self.transform is not None else out

    def check_complete(self, files):
        return self.file_name in files


class PartialPointCloudField(Field):
    """Surface samples cut by a random axis-aligned slab (fields.py:366-423)."""

    def __init__(self, file_name, transform=None, multi_files=None, part_ratio=0.7):
        self.file_name, self.transform = file_name, transform
        self.multi_files, self.part_ratio = multi_files, part_ratio

    def load(self, model_path, idx, category):
        z = np.load(_pick_file(model_path, self.file_name, self.multi_files), allow_pickle=True)
        points, normals = z['points'].astype(np.float32), z['normals'].astype(np.float32)
        side = np.random.randint(3)
        lo, hi = points[:, side].min(), points[:, side].max()
        length = np.random.uniform(self.part_ratio * (hi - lo), (hi - lo))
        keep = (points[:, side] - lo) <= length
        out = {None: points[keep], 'normals': normals[keep]}
        return self.transform(out) if self.transform is not None else out

    def check_complete(self, files):
        return self.file_name in files
