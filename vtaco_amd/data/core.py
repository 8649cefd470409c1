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


def _subdirs(folder):
    return [d for d in os.listdir(folder) if d and os.path.isdir(os.path.join(folder, d))]


def _category_table(root, categories):
    """{category: {'id', 'name', 'idx'}}: metadata.yaml where the dataset has one, placeholders otherwise; ``idx`` is the
    category's position in ``categories`` (what the fields receive as their third argument)."""
    meta_file = os.path.join(root, 'metadata.yaml')
    if os.path.exists(meta_file):
        with open(meta_file) as fh:
            table = yaml.safe_load(fh)
    else:
        table = {c: {'id': c, 'name': 'n/a'} for c in categories}
    for position, c in enumerate(categories):
        table[c]['idx'] = position
    return table


def _models_of(root, category, split):
    """Model names of one category: every sub-directory, or the lines of ``<split>.lst`` minus ONE empty entry (the reference
    removes exactly one, core.py:88-89; a file with several blank lines keeps the rest, and so does this)."""
    folder = os.path.join(root, category)
    if not os.path.isdir(folder):
        log.warning('Category %s does not exist in dataset.', category)
    if split is None:
        return _subdirs(folder)
    with open(os.path.join(folder, split + '.lst')) as fh:
        names = fh.read().split('\n')
    if '' in names:
        names.remove('')
    return names


def _flatten(name, value, sample):
    """A field's value into the flat sample: arrays as float32 under ``name`` (the unnamed entry) or ``name.key``; the model's
    name string stays a string."""
    if not isinstance(value, dict):
        sample[name] = value
        return
    for key, item in value.items():
        if key is None:
            sample[name] = item.astype(np.float32)
        else:
            sample[f'{name}.{key}'] = item if key == 'name' else item.astype(np.float32)


class Shapes3dDataset(tdata.Dataset):
    """``Shapes3dDataset(dataset_folder, fields, split, categories, no_except, transform, cfg)``: the reference's dataset class
    (src/data/core.py:34-183) over the layout in this module's header -- same constructor, same samples, same draw order.
    Crop / sliding-window mode (``input_type: pointcloud_crop``) is not built: no shipped VTacO config uses it."""

    def __init__(self, dataset_folder, fields, split=None, categories=None, no_except=True, transform=None, cfg=None):
        if cfg is not None and cfg.get('data', {}).get('input_type') == 'pointcloud_crop':
            raise NotImplementedError("Shapes3dDataset: input_type 'pointcloud_crop' is not built")
        self.dataset_folder, self.fields = dataset_folder, fields
        self.no_except, self.transform, self.cfg = no_except, transform, cfg
        categories = _subdirs(dataset_folder) if categories is None else categories
        self.metadata = _category_table(dataset_folder, categories)
        self.models = [{'category': c, 'model': m} for c in categories for m in _models_of(dataset_folder, c, split)]

    def __len__(self):
        return len(self.models)

    def __getitem__(self, idx):
        entry = self.models[idx]
        folder = os.path.join(self.dataset_folder, entry['category'], entry['model'])
        category_index = self.metadata[entry['category']]['idx']
        sample = {}
        for name, field in self.fields.items():
            try:
                _flatten(name, field.load(folder, idx, category_index), sample)
            except Exception:
                if not self.no_except:
                    raise
                # the reference swallows a broken sample and lets collate_remove_none drop it (core.py:154-164)
                log.warning('Error occured when loading field %s of model %s', name, entry['model'])
                return None
        return sample if self.transform is None else self.transform(sample)

    def get_model_dict(self, idx):
        return self.models[idx]

    def test_model_complete(self, category, model):
        folder = os.path.join(self.dataset_folder, category, model)
        present = os.listdir(folder)
        for name, field in self.fields.items():
            if not field.check_complete(present):
                log.warning('Field "%s" is incomplete: %s', name, folder)
                return False
        return True


def collate_remove_none(batch):
    """default_collate over the samples that loaded (core.py:255-264)."""
    return tdata.dataloader.default_collate([b for b in batch if b is not None])


def worker_init_fn(worker_id):
    """Fresh numpy seed per DataLoader worker (core.py:267-281)."""
    np.random.seed(int.from_bytes(os.urandom(4), byteorder='big') + worker_id)
