"""Dataset factory (reference src/config.py:121-219): ``get_dataset(mode, cfg)`` builds the
``Shapes3dDataset`` of a split from the same cfg keys (data.path, data.classes, data.*_split,
data.input_type, data.pointcloud_*, data.points_*)."""
from __future__ import annotations

from . import data
from .conv_onet import config as conv_onet_config

# every shipped config says ``method: vtaco`` (reference src/config.py:7-9); 'conv_onet' is kept as an alias
method_dict = {'vtaco': conv_onet_config, 'conv_onet': conv_onet_config}


class Compose:
    """torchvision.transforms.Compose without the dependency."""

    def __init__(self, transforms):
        self.transforms = list(transforms)

    def __call__(self, x):
        for t in self.transforms:
            x = t(x)
        return x


def get_inputs_field(mode, cfg):
    """Field of the network input (src/config.py:169-219)."""
    d = cfg['data']
    kind = d['input_type']
    if kind is None:
        return None
    if kind in ('pointcloud', 'partial_pointcloud'):
        transform = Compose([data.SubsamplePointcloud(d['pointcloud_n']), data.PointcloudNoise(d['pointcloud_noise'])])
        cls = data.PointCloudField if kind == 'pointcloud' else data.PartialPointCloudField
        return cls(d['pointcloud_file'], transform, multi_files=d['multi_files'])
    if kind == 'idx':
        return data.IndexField()
    if kind in ('pointcloud_crop', 'voxels'):
        raise NotImplementedError("get_inputs_field: input_type '%s' is not built (no shipped VTacO config uses it)" % kind)
    raise ValueError('Invalid input type (%s)' % kind)


def get_dataset(mode, cfg, return_idx=False):
    """Dataset of split ``mode`` in {'train','val','test'} (src/config.py:121-166)."""
    split = {'train': cfg['data']['train_split'], 'val': cfg['data']['val_split'], 'test': cfg['data']['test_split']}[mode]
    if cfg['data']['dataset'] != 'Shapes3D':
        raise ValueError('Invalid dataset "%s"' % cfg['data']['dataset'])
    fields = method_dict[cfg['method']].get_data_fields(mode, cfg)
    inputs = get_inputs_field(mode, cfg)
    if inputs is not None:
        fields['inputs'] = inputs
    if return_idx:
        fields['idx'] = data.IndexField()
    return data.Shapes3dDataset(cfg['data']['path'], fields, split=split, categories=cfg['data']['classes'], cfg=cfg)
