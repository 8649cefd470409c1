"""Model / generator factories (drop-in for the hot-path part of reference
src/conv_onet/config.py:16-143, 215-269): same cfg keys, same registries."""
from __future__ import annotations

from . import models
from .generation import Generator3D
from ..encoder import encoder_dict
from .._lib import VtError


def get_model(cfg, device=None, dataset=None, **kwargs):
    m = cfg['model']
    dim, c_dim, padding = cfg['data']['dim'], m['c_dim'], cfg['data']['padding']
    decoder = None
    if m['decoder']:
        decoder = models.decoder_dict[m['decoder']](
            dim=dim, c_dim=c_dim, padding=padding, with_contact=m.get('with_contact', False), **(m.get('decoder_kwargs') or {}))
    encoder = None
    if m['encoder']:
        encoder = encoder_dict[m['encoder']](dim=dim, c_dim=c_dim, padding=padding, **(m.get('encoder_kwargs') or {}))
    encoder_hand = None
    if m.get('encoder_hand'):
        # reference config.py:104-110: same registry, the hand encoder's own kwargs (planes, 2-D U-Net, MANO head)
        encoder_hand = encoder_dict[m['encoder_hand']](dim=dim, c_dim=c_dim, padding=padding,
                                                        **(m.get('encoder_hand_kwargs') or {}))
    encoder_img = None
    if m.get('with_img') and m.get('encoder_img'):
        encoder_img = encoder_dict[m['encoder_img']](**(m.get('encoder_img_kwargs') or {}))
    encoder_t2d = None
    if m.get('encoder_t2d'):
        kw = m['encoder_t2d_kwargs']
        img_t2d = encoder_dict[kw['encoder_img']](**kw['encoder_img_kwargs'])
        hand_t2d = None
        if kw.get('encoder_hand'):
            # the digit-pose regressor of the t2d model (config.py:126-131): c_dim comes from its own kwargs
            hand_t2d = encoder_dict[kw['encoder_hand']](dim=dim, padding=padding, **(kw.get('encoder_hand_kwargs') or {}))
        encoder_t2d = models.ConvolutionalOccupancyNetwork(None, None, hand_t2d, img_t2d, None, device=device)
        if kw.get('pretrained'):
            # reference config.py:131-133: CheckpointIO(out_dir, model=encoder_t2d).load(model_file) -- the file's 'model' entry
            # is the t2d net's state_dict (checkpoints.py:37-40, 91-93, strict).  A pretrained t2d net that was never loaded
            # would be neither trained (get_trainer drops its losses) nor meaningful, so a missing file raises like the reference
            load_t2d_checkpoint(encoder_t2d, kw.get('model_file'), (cfg.get('training') or {}).get('out_dir', '.'), device)
    return models.ConvolutionalOccupancyNetwork(decoder, encoder, encoder_hand, encoder_img, encoder_t2d, device=device)


def load_t2d_checkpoint(net, model_file, out_dir, device=None):
    """``CheckpointIO.load_file`` for the one module the factory loads (checkpoints.py:53-70, 84-98): relative paths are
    relative to ``training.out_dir``; the checkpoint is a dict whose ``'model'`` entry is the state_dict."""
    import os
    import torch
    if not model_file:
        raise VtError("get_model: encoder_t2d_kwargs.pretrained is true but no model_file is given")
    path = model_file if os.path.isabs(model_file) else os.path.join(out_dir, model_file)
    if not os.path.exists(path):
        raise FileNotFoundError(path)
    state = torch.load(path, map_location=device if device is not None else 'cpu')
    if 'model' not in state:
        raise VtError(f"get_model: {path} has no 'model' entry (keys: {sorted(state)[:8]})")
    net.load_state_dict(state['model'])


def get_generator(model, cfg, device, **kwargs):
    g = cfg['generation']
    return Generator3D(model, device=device, threshold=cfg['test']['threshold'], resolution0=g['resolution_0'],
                       upsampling_steps=g['upsampling_steps'], sample=g.get('use_sampling', False),
                       refinement_step=g.get('refinement_step', 0), simplify_nfaces=g.get('simplify_nfaces'),
                       input_type=cfg['data']['input_type'], padding=cfg['data']['padding'],
                       with_img=cfg['model'].get('with_img', False), encode_t2d=cfg['model'].get('encoder_t2d', False))


def get_trainer(model, optimizer, cfg, device, **kwargs):
    """reference conv_onet/config.py:146-212."""
    from .training import Trainer
    return Trainer(model, optimizer, device=device, input_type=cfg['data']['input_type'],
                   threshold=cfg['test']['threshold'], num_sample=cfg['data'].get('num_sample', 2048),
                   with_img=cfg['model'].get('with_img', False),
                   with_contact=cfg['model'].get('with_contact', False), train_tactile=cfg['model'].get('train_tactile', False),
                   encode_t2d=bool(cfg['model'].get('encoder_t2d', False)),
                   pretrained_t2d=(cfg['model'].get('encoder_t2d_kwargs') or {}).get('pretrained', True),
                   depth_origin=kwargs.get('depth_origin'))


def get_data_fields(mode, cfg):
    """Method-specific fields of a sample (reference conv_onet/config.py:272-318): the query points
    with occupancies ('points'), and for val / test the IoU points ('points_iou')."""
    from .. import data
    d = cfg['data']
    if d['input_type'] == 'pointcloud_crop':
        raise VtError("get_data_fields: crop / sliding-window mode is not built (no shipped config uses it)")
    fields = {}
    if d['points_file'] is not None:
        fields['points'] = data.PointsField(d['points_file'], data.SubsamplePoints(d['points_subsample']),
                                            unpackbits=d['points_unpackbits'], multi_files=d['multi_files'])
    if mode in ('val', 'test', 'vis'):
        if d['points_iou_file'] is not None:
            fields['points_iou'] = data.PointsField(d['points_iou_file'], unpackbits=d['points_unpackbits'],
                                                    multi_files=d['multi_files'])
        if d.get('voxels_file') is not None:
            raise VtError("get_data_fields: voxel (.binvox) fields are not built; set data.voxels_file: null")
    return fields
