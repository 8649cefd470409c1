"""Data formats either side of the hot path (SURVEY.md section 8f "next" row 4): the reference's dataset
layout, fields and transforms as host-side numpy, and the dataset factory of src/config.py."""
from .core import Field, Shapes3dDataset, collate_remove_none, worker_init_fn
from .fields import IndexField, PartialPointCloudField, PointCloudField, PointsField
from .meshes import load_mesh_dict, read_triangle_mesh
from .transforms import PointcloudNoise, SubsamplePointcloud, SubsamplePoints

__all__ = ["load_mesh_dict", "read_triangle_mesh", "Field", "Shapes3dDataset", "collate_remove_none", "worker_init_fn", "IndexField", "PointsField",
           "PointCloudField", "PartialPointCloudField", "PointcloudNoise", "SubsamplePointcloud", "SubsamplePoints"]
