"""Oracle: kNN-PCA normal estimation + sensor orientation (TEST INFRASTRUCTURE ONLY).

Restates the recipe the reference ships in source form, examples/recons_waymo_cpu.py:21-41 (the CPU stand-in
for ``nksr.get_estimate_normal_preprocess_fn(64, 85.0)``, examples/recons_waymo.py:36), with scipy's cKDTree +
numpy's eigh in place of ``point_cloud_utils.estimate_point_cloud_normals_knn`` (:26, not installed):
  :26     unoriented normal = eigenvector of the smallest eigenvalue of the covariance of the k nearest
          neighbours (the point itself included)
  :30-36  flip towards the sensor: view = (sensor - xyz) / (|sensor - xyz| + 1e-6);  n <- -n where view.n < 0
  :38-39  keep |view.n| > cos(deg)
Independent of nksr_amd/normals.py (exact kd-tree kNN, LAPACK eigen-solve, fp64 covariance).
"""
import os

import numpy as np


def estimate_normals_knn(xyz, sensor, knn=64, deg=85.0, workers=None):
    """(xyz [N,3], sensor [N,3]) -> (xyz' [N',3], normal' [N',3], keep mask [N], cos [N]); input order kept."""
    from scipy.spatial import cKDTree
    xyz = np.asarray(xyz, np.float32)
    if xyz.shape[0] < knn:
        raise RuntimeError('need at least knn=%d points' % knn)
    _, nb = cKDTree(xyz).query(xyz, k=knn, workers=workers if workers is not None else (os.cpu_count() or 1))
    p = xyz[nb].astype(np.float64)
    d = p - p.mean(1, keepdims=True)
    cov = np.einsum('nki,nkj->nij', d, d)
    _, v = np.linalg.eigh(cov)
    nrm = v[:, :, 0].astype(np.float32)
    view = np.asarray(sensor, np.float32) - xyz
    view = view / (np.linalg.norm(view, axis=-1, keepdims=True) + np.float32(1e-6))
    cos = (view * nrm).sum(1)
    nrm[cos < 0] *= -1
    keep = np.abs(cos) > np.cos(np.deg2rad(deg))
    return xyz[keep], nrm[keep], keep, cos
