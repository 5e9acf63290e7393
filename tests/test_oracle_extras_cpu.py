"""CPU tests of the oracle pieces added for the chunked / sensor-only configurations and the quality pins
(oracle/chunking.py, oracle/normals.py, oracle/metrics.py, oracle/waymo_cpu.py) and of the committed fixtures'
internal consistency.  Sized to run in well under a minute."""
import json
import os
import subprocess
import sys

import numpy as np

import parity_util as pu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden')


def _two_spheres(n=2500):
    from nksr_amd import utils
    a, na = utils.synth_sphere(n, 0.7, 0.0, seed=1, center=(-0.9, 0.0, 0.0))
    b, nb = utils.synth_sphere(n, 0.7, 0.0, seed=2, center=(0.9, 0.15, 0.0))
    xyz = np.concatenate([a, b]).astype(np.float32)
    return xyz - xyz.min(0), np.concatenate([na, nb]).astype(np.float32)


def test_oracle_chunk_blend_is_a_partition_of_unity_and_closes_the_seam():
    from oracle import chunking, pipeline
    xyz, nrm = _two_spheres()
    ext = float(xyz[:, 0].max())
    cf = chunking.reconstruct_by_chunk(xyz, nrm, None, ext / 2 + 1e-3, tol=1e-6)
    assert cf.grid == [2, 1, 1] and sorted(cf.fields) == [0, 1]
    # weights: the normalised weight of chunk 0 falls monotonically from 1 to 0 across the seam; a chunk weighs nothing
    # beyond ov outside its core; the sum never vanishes inside the scene
    x = np.linspace(0, ext, 400).astype(np.float32)
    q = np.stack([x, np.full_like(x, 0.5), np.full_like(x, 0.5)], 1)
    w0, w1 = cf.weight(0, q), cf.weight(1, q)
    assert ((w0 + w1)[1:-1] > 0).all()
    r0 = (w0 / np.maximum(w0 + w1, 1e-20))[1:-1]
    assert (np.diff(r0) <= 1e-6).all() and r0[0] == 1.0 and r0[-1] == 0.0
    seam = cf.cores[0][1][0]
    assert (w1[x < seam - cf.ov - 1e-3] == 0).all() and (w0[x > seam + cf.ov + 1e-3] == 0).all()
    # the blended field matches each chunk's own field where only that chunk weighs
    f, _ = cf.evaluate(xyz)
    only0 = (cf.weight(1, xyz) == 0)
    f0, _ = pipeline.evaluate(cf.fields[0], (xyz[only0] + cf.frame.shift(0)[None]).astype(np.float32))      # chunk 0 lives in its slot of the exploded frame
    np.testing.assert_allclose(f[only0], f0, rtol=1e-6, atol=1e-9)
    v, t = cf.extract_dual_mesh(0)
    pu.assert_closed(t, 'chunked oracle mesh')
    e = np.sort(np.concatenate([t[:, [0, 1]], t[:, [1, 2]], t[:, [2, 0]]]), 1)
    assert len(v) - len(np.unique(e, axis=0)) + len(t) == 4          # two spheres
    # the adaptive dual graph of the UNION hierarchy (one level here: the lattice mesher's cells, same surface)
    va, ta = cf.extract_dual_mesh(0, dual_graph='adaptive')
    pu.assert_closed(ta, 'chunked oracle mesh, adaptive dual graph')
    assert len(ta) == len(t)


def test_oracle_normals_recipe_on_a_sphere():
    from nksr_amd import utils
    from oracle import normals
    xyz, radial = utils.synth_sphere(4000, 1.0, 0.0, seed=3)
    sensor = np.zeros_like(xyz)                                       # scanner at the centre: normals must point inward
    xs, ns, keep, cos = normals.estimate_normals_knn(xyz, sensor, 32, 85.0, workers=1)
    assert keep.all() and len(xs) == len(xyz)
    assert ((ns * radial).sum(1) < -0.99).all()
    far = np.tile(np.array([[10.0, 0, 0]], np.float32), (len(xyz), 1))    # a distant scanner: grazing points are dropped
    xs, ns, keep, cos = normals.estimate_normals_knn(xyz, far, 32, 85.0, workers=1)
    assert 0.8 < keep.mean() < 0.99
    view = far[keep] - xs
    assert ((view * ns).sum(1) > 0).all()


def test_oracle_metrics_known_answers():
    from oracle import metrics
    # unit square in z=0 vs the same square shifted by d: chamfer-L1 == d (interior), normals consistent
    v = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]], np.float64)
    f = np.array([[0, 1, 2], [0, 2, 3]])
    p, n = metrics.sample_mesh(v, f, 20000, seed=0)
    assert np.allclose(p[:, 2], 0) and np.allclose(np.abs(n[:, 2]), 1) and 0.45 < p[:, 0].mean() < 0.55
    gt = p.copy()
    gt[:, 2] += 0.005
    m = metrics.evaluate(p, n, gt, n)
    assert abs(m['chamfer-L1'] - 0.005) < 2e-4 and m['f-score'] == 1.0 and m['normals'] > 0.999
    gt[:, 2] += 0.02
    m = metrics.evaluate(p, n, gt, n)
    assert m['f-score'] == 0.0 and m['f-score-outdoor'] == 1.0


def test_waymo_cpu_worker_runs_the_named_sequence():
    """bench.py's cpu_baseline leg: one worker process, one run of the examples/recons_waymo_cpu.py sequence on the bunny."""
    env = dict(os.environ, OMP_NUM_THREADS='1')
    r = subprocess.run([sys.executable, '-m', 'oracle.waymo_cpu', '--worker', os.path.join(GOLD, 'bunny_10k.npz'), '1'],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out['points'] == 10000 and out['faces'] > 1000 and out['seconds'] > 0


def test_chunked_fixtures_are_self_consistent():
    """tests/golden/{street8,terrain5}_golden.npz: mesh arrays agree with each other and with the declared grids."""
    for name, grid, nch in (('street8', [4, 2, 1], 8), ('terrain5', [2, 2, 1], 4)):
        g = np.load(os.path.join(GOLD, name + '_golden.npz'))
        assert [int(v) for v in g['grid']] == grid and len(g['chunk_ids']) == nch
        ref = pu.ref_from_golden(g)
        assert ref['f'].max() == len(ref['v']) - 1 and len(ref['vert_vkey']) == len(ref['v']) == len(ref['vert_df'])
        tri = pu.canonical_triangles(ref['f'], ref['vert_vkey'], ref['vert_axis'])
        lo, hi = pu.triangle_cells(tri)
        assert (lo <= hi).all()
        # every mesh vertex lies on its lattice edge: lower end point + t * h along the axis, t in [0, 1]
        from oracle import meshing as om
        p0 = om.lattice_positions(om.lattice_decode(ref['vert_vkey']), ref['h'], 0.05)
        d = ref['v'] - p0
        ax = ref['vert_axis'].astype(np.int64)
        along = d[np.arange(len(d)), ax]
        assert (along >= -1e-6).all() and (along <= ref['h'] + 1e-6).all()
        d[np.arange(len(d)), ax] = 0
        assert np.abs(d).max() < 1e-6


def test_sdfgen_oracle_on_an_analytic_sphere():
    """oracle/sdfgen.py (restatement of ext/sdfgen/sdf_from_points.cu): on a dense noise-free sphere every estimator returns the
    radial offset of the query; the exact kNN it uses is checked against brute force."""
    from oracle import sdfgen
    rs = np.random.RandomState(0)
    v = rs.randn(4000, 3)
    nrm = (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)
    xyz = (nrm * 0.5).astype(np.float32)
    u = rs.randn(200, 3)
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    off = rs.uniform(-0.03, 0.03, 200)
    q = (u * (0.5 + off)[:, None]).astype(np.float32)
    for kw in (dict(nb_points=8, stdv=0.02), dict(nb_points=8, stdv=3.0, adaptive_knn=8), dict(nb_points=8, stdv=0.05, imls=True)):
        s, g = sdfgen.sdf_from_points(q, xyz, nrm, compute_grad=True, **kw)
        # IMLS / in-threshold votes project on the neighbours' tangent planes; a vote outside stdv * ref_std returns the distance to the
        # nearest SAMPLE (spacing ~0.03 here), which over-estimates the radial offset
        assert np.abs(s - off).max() < (4e-3 if kw.get('imls') or kw.get('adaptive_knn') else 0.035), (kw, np.abs(s - off).max())
        assert (np.sign(s) == np.sign(off))[np.abs(off) > 2e-3].all()
        if kw.get('imls') or kw.get('adaptive_knn'):
            assert (np.sum(g * u, axis=1) > 0.95)[np.abs(off) > 2e-3].all()
    # brute-force neighbours of a few queries
    d2 = ((q[:20, None, :].astype(np.float64) - xyz[None].astype(np.float64)) ** 2).sum(-1)
    idx = np.argsort(d2, axis=1)[:, :8]
    ray = q[:20, None, :] - xyz[idx]
    d = (nrm[idx] * ray).sum(-1)
    e = (ray ** 2).sum(-1) / 0.05 ** 2
    w = np.exp(-e + e.min(1, keepdims=True))
    ref = (d * w).sum(1) / w.sum(1)
    s, = sdfgen.sdf_from_points(q[:20], xyz, nrm, 8, 0.05, imls=True)
    assert np.abs(s - ref).max() < 1e-6
