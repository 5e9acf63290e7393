"""Wall time of the non-headline BASELINE.json configurations (parity-test cases, not bench lines).
python -m nksr_amd.tools.prof_configs"""
import time

import numpy as np
import torch

import nksr_amd
from nksr_amd import configs, utils


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, out


def main():
    dev = torch.device('cuda:0')
    t = lambda a: torch.from_numpy(a).to(dev)
    # configs[1]: ShapeNet 3K-noise stand-in (kernel_dim 16, interpolator hidden 32, voxel 0.02)
    rec = nksr_amd.Reconstructor(dev, config='snet-n3k-wnormal')
    for n in (3000, 100000):
        xyz, nrm = utils.synth_sphere(n, 0.45, 0.005, seed=0)
        dt, mesh = timed(lambda: rec.reconstruct(t(xyz), t(nrm)).extract_dual_mesh(mise_iter=1))
        print('configs[1] snet-n3k-wnormal  N=%-7d  %.1f ms  (%.2f M points/s)  F=%d' % (n, dt * 1e3, n / dt / 1e6, mesh.f.shape[0]))
    # configs[3]: CARLA-like street, sensor-only input, kNN normals, 8 chunks, UDF mask (carla preset)
    rec = nksr_amd.Reconstructor(dev, config='carla')
    xyz, nrm = utils.synth_terrain(400000, seed=1, extent=(200.0, 100.0))
    sensor = np.tile(np.array([[100.0, 50.0, 30.0]], np.float32), (xyz.shape[0], 1))
    fn = nksr_amd.get_estimate_normal_preprocess_fn(64, 85.0)
    scale = 0.1 / 0.5                                     # 0.5 m voxels
    dt, mesh = timed(lambda: rec.reconstruct(t(xyz * np.float32(scale)), sensor=t(sensor * np.float32(scale)), detail_level=None,
                                              chunk_size=50.0 * scale, preprocess_fn=fn).extract_dual_mesh(mise_iter=1), reps=2)
    print('configs[3] carla preset, sensor-only 400k points, 4x2 chunks of 50 m  %.1f ms  (%.2f M points/s)  F=%d' % (
        dt * 1e3, xyz.shape[0] / dt / 1e6, mesh.f.shape[0]))


if __name__ == '__main__':
    main()
