"""Oracle: the reference's CPU example call sequence, timed on the host cores (TEST INFRASTRUCTURE / bench.py's
``cpu_baseline`` leg only).

BASELINE.json north_star: "the reference's examples/recons_waymo_cpu.py timed on the host cores (count stated) in the same
run as the reported baseline"; configs[0]: "examples/recons_waymo_cpu.py on assets/bunny.ply, tree_depth=4, CPU".
The script needs the absent ``nksr`` wheel, ``point_cloud_utils`` and a downloaded Waymo cloud, so
what runs here is its CALL SEQUENCE on the CPU restatement (kind "port"):
    examples/recons_waymo_cpu.py:21-41   normal_func: kNN-64 PCA normals, flip to the sensor, drop > 85 degrees
    :55-61  reconstruct(xyz, sensor=..., detail_level=None, approx_kernel_grad=True, solver_tol=1e-4, fused_mode=True,
                        preprocess_fn=normal_func)
    :63     field.extract_dual_mesh(mise_iter=1)
on ``assets/bunny.ply`` (committed as tests/golden/bunny_10k.npz: the asset is test input, and /root/reference does
not exist on the GPU box).  The asset has no scanner positions: six scanners on the axes around the shape are
synthesised and every point gets the one that faces it.
The numpy/scipy restatement is single-threaded apart from the kd-tree query, so "all host cores" is realised as
``cores`` concurrent worker processes, each running the whole sequence (throughput of a batch of scans -- the
way a CPU node would serve independent requests or chunks); the single-process rate is reported next to it.

    python -m oracle.waymo_cpu --worker <input.npz> <repeats> [crop.npz]     (one worker; prints a JSON line)
"""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUNNY = os.path.join(ROOT, 'tests', 'golden', 'bunny_10k.npz')


def synth_sensors(xyz, normal, dist=2.0):
    """Six scanner positions on the axes around the centroid; every point is seen from the one its normal faces."""
    c = xyz.mean(0)
    S = (np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], np.float32) * np.float32(dist) + c).astype(np.float32)
    d = S[None] - xyz[:, None]
    d /= np.linalg.norm(d, axis=2, keepdims=True)
    return S[(d * normal[:, None]).sum(2).argmax(1)]


def _interps(depth, K=4, H=16):
    """Untrained kitchen-sink interpolators: zero last layer (phi == t), identical to
    oracle.pipeline.default_interpolators(init_scale=0) in effect, without importing torch in the workers."""
    from . import kernel
    rs = np.random.RandomState(0)
    return [kernel.Interpolator(rs.randn(H, K) / K ** 0.5, np.zeros(H), rs.randn(H, H) / H ** 0.5, np.zeros(H), np.zeros((K, H)), np.zeros(K))
            for _ in range(depth)]


def run_sequence(xyz, sensor, knn=64, deg=85.0, workers=1, timing=None):
    """examples/recons_waymo_cpu.py:48-63 on the oracle.  Returns (n_input, mesh vertices, mesh faces)."""
    from . import normals, pipeline
    xs, ns, _, _ = normals.estimate_normals_knn(xyz, sensor, knn, deg, workers=workers)
    fld = pipeline.reconstruct(xs, ns, approx_kernel_grad=True, tol=1e-4, interps=_interps(4), timing=timing)
    v, f = pipeline.extract_dual_mesh(fld, mise_iter=1)
    return xyz.shape[0], v, f


def run_oriented(xyz, normal, mise_iter=1, tol=1e-5):
    from . import pipeline
    fld = pipeline.reconstruct(xyz, normal, tol=tol, interps=_interps(4))
    v, f = pipeline.extract_dual_mesh(fld, mise_iter=mise_iter)
    return xyz.shape[0], v, f


def _worker(argv):
    d = np.load(argv[0])
    repeats = int(argv[1])
    xyz, nrm = d['xyz'], d['normal']
    sensor = synth_sensors(xyz, nrm)
    t0 = time.perf_counter()
    for _ in range(repeats):
        n, v, f = run_sequence(xyz, sensor)
    out = {'points': int(n) * repeats, 'seconds': time.perf_counter() - t0, 'faces': int(len(f))}
    if len(argv) > 2:
        c = np.load(argv[2])
        t0 = time.perf_counter()
        n, v, f = run_oriented(c['xyz'], c['normal'], int(c['mise_iter']))
        out.update({'crop_points': int(n), 'crop_seconds': time.perf_counter() - t0, 'crop_faces': int(len(f))})
    print(json.dumps(out))


def measure(cores=None, repeats=4, crop=None, timeout=600):
    """Launches ``cores`` concurrent workers (fresh interpreters: no torch, no HIP).  ``crop``: optional path of an
    .npz (xyz in model units, normal, mise_iter) -- a bounded sample of the bench workload, one per worker run.
    Returns the ``cpu_baseline`` record."""
    cores = cores or os.cpu_count() or 1
    env = dict(os.environ, OMP_NUM_THREADS='1', OPENBLAS_NUM_THREADS='1', MKL_NUM_THREADS='1')
    cmd = [sys.executable, '-m', 'oracle.waymo_cpu', '--worker', BUNNY, str(repeats)] + ([crop] if crop else [])
    t0 = time.perf_counter()
    procs = [subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for _ in range(cores)]
    res = []
    for p in procs:
        so, se = p.communicate(timeout=timeout)
        if p.returncode != 0:
            raise RuntimeError('cpu baseline worker failed: %s' % se[-2000:])
        res.append(json.loads(so.strip().splitlines()[-1]))
    wall = time.perf_counter() - t0
    pts = sum(r['points'] for r in res)
    slow = max(r['seconds'] for r in res)
    rec = {'value': pts / slow, 'unit': 'points/s', 'cores': cores, 'kind': 'port',
           'single_process_value': res[0]['points'] / res[0]['seconds'] if cores == 1 else None,
           'sample': 'examples/recons_waymo_cpu.py:48-63 call sequence (kNN-64 normals from synthesised scanners, approx_kernel_grad, '
                     'solver_tol 1e-4, extract_dual_mesh(mise_iter=1)) on assets/bunny.ply (10 000 points), oracle port; %d concurrent '
                     'single-threaded worker processes x %d runs each, slowest worker %.1f s (wall incl. start-up %.1f s)' % (
                         cores, repeats, slow, wall)}
    if crop:
        cs = max(r['crop_seconds'] for r in res)
        rec['workload_crop'] = {'value': sum(r['crop_points'] for r in res) / cs, 'unit': 'points/s',
                                'sample': 'oracle reconstruct+extract_dual_mesh on a %d-point spatial crop of the bench cloud per worker, %d '
                                          'concurrent workers, slowest %.1f s' % (res[0]['crop_points'], cores, cs)}
    return rec


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == '--worker':
        _worker(sys.argv[2:])
    else:
        print(json.dumps(measure(repeats=int(sys.argv[1]) if len(sys.argv) > 1 else 2)))
