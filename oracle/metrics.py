"""Oracle: mesh quality metrics (TEST INFRASTRUCTURE ONLY).

Restates the reference's ``MeshEvaluator`` (metrics.py:46-192) with scipy's cKDTree in place of pykdtree and a
numpy area-weighted sampler in place of open3d's ``sample_points_uniformly`` (metrics.py:95-99):
  distance_p2p            metrics.py:19-36   nearest-neighbour distance + |normal dot product|
  completeness / accuracy metrics.py:121-146 mean distance gt->pd / pd->gt
  chamfer-L1 / -L2        metrics.py:148-153 mean of the two (squared for L2)
  f-score                 metrics.py:155-159 harmonic mean of precision / recall at 0.01 (0.015, 0.02; 0.1 outdoor)
This is the reference-independent quality pin of SURVEY.md section 8c(3): HIP meshes of analytic shapes are
scored against dense samples of the analytic surface.
"""
import numpy as np

THRESHOLDS = np.array([0.01, 0.015, 0.02, 0.002, 0.1])      # metrics.py:72


def sample_mesh(v, f, n, seed=0):
    """n area-uniform samples + the (unit) triangle normals they lie on (metrics.py:95-99)."""
    v = np.asarray(v, np.float64)
    f = np.asarray(f, np.int64)
    if len(f) == 0:
        return np.zeros((0, 3)), np.zeros((0, 3))
    a, b, c = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    cr = np.cross(b - a, c - a)
    area = 0.5 * np.linalg.norm(cr, axis=1)
    rs = np.random.RandomState(seed)
    t = rs.choice(len(f), size=n, p=area / area.sum())
    r1, r2 = np.sqrt(rs.rand(n)), rs.rand(n)
    p = (1 - r1)[:, None] * a[t] + (r1 * (1 - r2))[:, None] * b[t] + (r1 * r2)[:, None] * c[t]
    nrm = cr[t] / np.maximum(np.linalg.norm(cr[t], axis=1, keepdims=True), 1e-30)
    return p, nrm


def distance_p2p(src, nsrc, tgt, ntgt):
    from scipy.spatial import cKDTree
    dist, idx = cKDTree(tgt).query(src)
    dots = None
    if nsrc is not None and ntgt is not None:
        a = nsrc / np.maximum(np.linalg.norm(nsrc, axis=-1, keepdims=True), 1e-30)
        b = ntgt / np.maximum(np.linalg.norm(ntgt, axis=-1, keepdims=True), 1e-30)
        dots = np.abs((b[idx] * a).sum(-1))          # orientation-agnostic, metrics.py:30-32
    return dist, dots


def evaluate(pd, pd_n, gt, gt_n):
    """Metric dict of a predicted sample set against ground-truth samples (metrics.py:108-178)."""
    comp, comp_n = distance_p2p(gt, gt_n, pd, pd_n)
    acc, acc_n = distance_p2p(pd, pd_n, gt, gt_n)
    recall = [(comp <= t).mean() for t in THRESHOLDS]
    precision = [(acc <= t).mean() for t in THRESHOLDS]
    F = [2 * p * r / max(p + r, 1e-30) for p, r in zip(precision, recall)]
    out = {'completeness': comp.mean(), 'accuracy': acc.mean(), 'chamfer-L1': 0.5 * (comp.mean() + acc.mean()),
           'chamfer-L2': 0.5 * ((comp ** 2).mean() + (acc ** 2).mean()), 'f-precision': precision[0], 'f-recall': recall[0],
           'f-score': F[0], 'f-score-15': F[1], 'f-score-20': F[2], 'f-score-outdoor': F[4]}
    if comp_n is not None:
        out['normals'] = 0.5 * (comp_n.mean() + acc_n.mean())
    return {k: float(v) for k, v in out.items()}


def eval_mesh(v, f, gt, gt_n, n_points=100000, seed=0):
    p, n = sample_mesh(v, f, n_points, seed)
    if len(p) == 0:
        return {k: float('nan') for k in ('chamfer-L1', 'f-score', 'normals')}
    return evaluate(p, n, np.asarray(gt, np.float64), None if gt_n is None else np.asarray(gt_n, np.float64))
