"""Asset loaders for the example scripts.  The reference's examples/common.py downloads its clouds
(examples/common.py:34-74) and reads them with pycg / pyntcloud; there is no network here, so the
loaders fall back to deterministic synthetic clouds when the reference assets are not on disk."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nksr_amd import utils  # noqa: E402

REF_ASSETS = os.environ.get('NKSR_ASSETS', '/root/reference/assets')


def load_bunny_example():
    p = os.path.join(REF_ASSETS, 'bunny.ply')
    if os.path.exists(p):
        xyz, nrm, _, _ = utils.load_point_cloud(p)
        return xyz, nrm
    return utils.synth_torus(10000, 0.32, 0.12, 0.0, seed=0)


def load_spot_example():
    p = os.path.join(REF_ASSETS, 'spot.ply')
    if os.path.exists(p):
        xyz, nrm, col, _ = utils.load_point_cloud(p)
        return xyz, nrm, col
    xyz, nrm = utils.synth_sphere(20000, 0.45, 0.0, seed=0)
    return xyz, nrm, (nrm * 0.5 + 0.5).astype(np.float32)


def load_buda_example(n=2_000_000):
    """Stand-in for the downloadable 'buda' scan: a large multi-object scene."""
    return utils.synth_scene(n, seed=0, extent=(120.0, 80.0, 10.0), noise=0.01, n_objects=24)


def load_waymo_example(n=1_500_000):
    """Stand-in for the Waymo sequence: terrain patch + sensor positions along a polyline."""
    xyz, nrm = utils.synth_terrain(n, seed=0, extent=(200.0, 100.0))
    t = np.clip((xyz[:, 0] - xyz[:, 0].min()) / 200.0, 0, 1)
    sensor = np.stack([xyz[:, 0].min() + 200.0 * t, np.full(n, 50.0), np.full(n, 25.0)], 1).astype(np.float32)
    return xyz, sensor


def warning_on_low_memory(threshold_mb):
    utils.warning_on_low_memory(threshold_mb)


def load_scannet_example(n=400_000):
    """Stand-in for the downloadable ScanNet room (examples/common.py of the reference): an indoor-sized scene in metres -- a few
    objects in an 8 x 6 x 3 m box, oriented, 5 mm noise -- so that voxel_size=0.02 (examples/recons_scannet.py:28) means what it
    means there."""
    return utils.synth_scene(n, seed=1, extent=(8.0, 6.0, 3.0), noise=0.005, n_objects=6)


def load_las_example(n=300_000):
    """Stand-in for the aerial LAS tile of examples/gis_app.py:14-30: terrain in metres with a large coordinate offset (the
    example subtracts the mean before reconstruction and adds it back to mesh.v), seen from one sensor high above."""
    xyz, _ = utils.synth_terrain(n, seed=2, extent=(40.0, 40.0))
    offset = np.array([433_200.0, 5_213_400.0, 310.0])
    return xyz.astype(np.float64) + offset[None], offset
