"""Dataset side of the path's callers (SURVEY.md section 8f "next" row 4): the on-disk layout the
reference trains and evaluates from, read into the dictionaries ``Trainer`` / ``Generator3D`` consume.

Layout (reference src/data/core.py:34-141, train.py:63-66)::

    <dataset>/metadata.yaml                     optional {category: {id, name}}
    <dataset>/<category>/<split>.lst            model names, one per line
    <dataset>/<category>/<model>/points.npz     PointsField      (fields.py)
    <dataset>/<category>/<model>/pointcloud.npz PointCloudField  (fields.py)

Host-side numpy: nothing here touches the GPU.  The random draws (file choice, shuffles,
subsampling, noise) go through ``np.random`` in the reference's order, so a seeded run sees the
same samples (tests/test_data_cpu.py pins that against the reference's own loader).
"""
from __future__ import annotations

import logging
import os

import numpy as np
import yaml
from torch.utils import data as tdata

log = logging.getLogger(__name__)


class Field:
    """One entry of a sample.  ``load`` returns a value or a dict {None: main array, key: extra}."""

    def load(self, model_path, idx, category):
        raise NotImplementedError

    def check_complete(self, files):
        raise NotImplementedError


class Shapes3dDataset(tdata.Dataset):
    """``Shapes3dDataset(dataset_folder, fields, split, categories, no_except, transform, cfg)``
    (core.py:34-141).  Crop / sliding-window mode (``input_type: pointcloud_crop``) is not built: no
    shipped VTacO config uses it."""

    def __init__(self, dataset_folder, fields, split=None, categories=None, no_except=True, transform=None, cfg=None):
        self.dataset_folder, self.fields = dataset_folder, fields
        self.no_except, self.transform, self.cfg = no_except, transform, cfg
        if cfg is not None and cfg.get('data', {}).get('input_type') == 'pointcloud_crop':
            raise NotImplementedError("Shapes3dDataset: input_type 'pointcloud_crop' is not built")
        if categories is None:
            categories = [c for c in os.listdir(dataset_folder) if os.path.isdir(os.path.join(dataset_folder, c))]
        meta_path = os.path.join(dataset_folder, 'metadata.yaml')
        if os.path.exists(meta_path):
            with open(meta_path) as fh:
                self.metadata = yaml.safe_load(fh)
        else:
            self.metadata = {c: {'id': c, 'name': 'n/a'} for c in categories}
        for c_idx, c in enumerate(categories):
            self.metadata[c]['idx'] = c_idx
        self.models = []
        for c in categories:
            sub = os.path.join(dataset_folder, c)
            if not os.path.isdir(sub):
                log.warning('Category %s does not exist in dataset.', c)
            if split is None:
                names = [d for d in os.listdir(sub) if d != '' and os.path.isdir(os.path.join(sub, d))]
            else:
                with open(os.path.join(sub, split + '.lst')) as fh:
                    names = fh.read().split('\n')
                if '' in names:
                    names.remove('')                  # the reference drops ONE empty entry (core.py:88-89)
            self.models += [{'category': c, 'model': m} for m in names]

    def __len__(self):
        return len(self.models)

    def __getitem__(self, idx):
        category, model = self.models[idx]['category'], self.models[idx]['model']
        c_idx = self.metadata[category]['idx']
        model_path = os.path.join(self.dataset_folder, category, model)
        sample = {}
        for name, field in self.fields.items():
            try:
                value = field.load(model_path, idx, c_idx)
            except Exception:
                if self.no_except:
                    log.warning('Error occured when loading field %s of model %s', name, model)
                    return None
                raise
            if isinstance(value, dict):
                for k, v in value.items():
                    if k is None:
                        sample[name] = v.astype(np.float32)
                    elif k == 'name':
                        sample['%s.%s' % (name, k)] = v
                    else:
                        sample['%s.%s' % (name, k)] = v.astype(np.float32)
            else:
                sample[name] = value
        if self.transform is not None:
            sample = self.transform(sample)
        return sample

    def get_model_dict(self, idx):
        return self.models[idx]

    def test_model_complete(self, category, model):
        files = os.listdir(os.path.join(self.dataset_folder, category, model))
        for name, field in self.fields.items():
            if not field.check_complete(files):
                log.warning('Field "%s" is incomplete: %s', name, os.path.join(self.dataset_folder, category, model))
                return False
        return True


def collate_remove_none(batch):
    """default_collate over the samples that loaded (core.py:255-264)."""
    return tdata.dataloader.default_collate([b for b in batch if b is not None])


def worker_init_fn(worker_id):
    """Fresh numpy seed per DataLoader worker (core.py:267-281)."""
    np.random.seed(int.from_bytes(os.urandom(4), byteorder='big') + worker_id)
