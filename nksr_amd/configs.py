"""Hyper-parameter sets and checkpoint loading -- mirror of ``nksr.configs``.

Values follow the reference's YAML (configs/default/train.yaml:9-29,
configs/shapenet/train_3k_noise.yaml:13-18, configs/carla/train.yaml:6).  The reference
downloads weights by URL (``load_checkpoint_from_url``, models/nksr_net.py:17,36-38); there
is no network here, so that entry point only accepts local files.
"""
import copy
import os

import torch


class HParams(dict):
    """dict with attribute access (the reference passes OmegaConf nodes)."""

    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError:
            raise AttributeError(k)
        return v

    def __setattr__(self, k, v):
        self[k] = v

    @staticmethod
    def wrap(d):
        out = HParams()
        for k, v in d.items():
            out[k] = HParams.wrap(v) if isinstance(v, dict) else v
        return out


_DEFAULT = {
    'feature': 'normal', 'geometry': 'kernel', 'voxel_size': 0.1, 'kernel_dim': 4, 'tree_depth': 4,
    'adaptive_depth': 1, 'unet': {'f_maps': 32}, 'udf': {'enabled': False},
    'interpolator': {'n_hidden': 2, 'hidden_dim': 16},
    'solver': {'pos_weight': 10000.0, 'normal_weight': 10000.0},
    'seed': 0, 'interpolator_init_scale': 0.0, 'head_init_scale': 0.0,
    # [ASSUMPTION] (the layer that makes feat.normal_features is in the absent wheel): the normal targets are the
    # splatted input normals divided by max(|nv|, normal_min_length * sum w, normal_min_weight) -- unit length unless the normals of a
    # voxel's points cancel or the voxel is barely touched.  0 / 0 = plain unit normalisation nv / |nv|.
    'normal_min_length': 1e-2, 'normal_min_weight': 1e-3,
}

_PRESETS = {
    'default': {},
    'ks': {},                                           # kitchen-sink: default architecture
    'snet-n3k-wnormal': {'voxel_size': 0.02, 'kernel_dim': 16, 'interpolator': {'n_hidden': 2, 'hidden_dim': 32}},
    'carla': {'adaptive_depth': 2, 'udf': {'enabled': True}},      # configs/carla/train.yaml:6-9
}


def get_hparams(name='ks', **overrides):
    if name not in _PRESETS:
        raise RuntimeError('unknown config %r (have %s)' % (name, sorted(_PRESETS)))
    d = copy.deepcopy(_DEFAULT)
    for src in (_PRESETS[name], overrides):
        for k, v in src.items():
            if isinstance(v, dict) and isinstance(d.get(k), dict):
                d[k].update(v)
            else:
                d[k] = v
    return HParams.wrap(d)


def load_checkpoint_from_url(url, map_location='cpu'):
    """Reference: downloads + caches a .pth and returns a dict with 'state_dict'
    (models/nksr_net.py:36-38).  Offline build: ``url`` must be a local path or file:// URL."""
    path = url[7:] if url.startswith('file://') else url
    if not os.path.exists(path):
        raise RuntimeError('no network access in this build: cannot fetch %r; pass a local .pth path' % url)
    ckpt = torch.load(path, map_location=map_location)
    return ckpt if 'state_dict' in ckpt else {'state_dict': ckpt}
