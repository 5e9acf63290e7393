import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


def make_cloud(kind='sphere', n=3000, noise=0.005, seed=0):
    """ShapeNet-3K-noise style inputs (dataset/transforms.py:34-48, configs/shapenet/train_3k_noise.yaml:4-8)
    on analytic shapes inside [-0.5, 0.5]^3."""
    from nksr_amd import utils
    if kind == 'sphere':
        xyz, nrm = utils.synth_sphere(n, 0.45, noise, seed)
    elif kind == 'torus':
        xyz, nrm = utils.synth_torus(n, 0.32, 0.12, noise, seed)
    elif kind == 'rbox':
        xyz, nrm = utils.synth_rounded_box(n, (0.30, 0.22, 0.16), 0.10, noise, seed)
    else:
        raise ValueError(kind)
    return xyz, nrm


@pytest.fixture(scope='session')
def bunny():
    from nksr_amd import utils
    path = os.path.join(ROOT, 'tests', 'golden', 'bunny_2k.npz')
    d = np.load(path)
    return d['xyz'], d['normal']
