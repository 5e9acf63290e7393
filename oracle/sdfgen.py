"""ORACLE (test infrastructure only -- never imported by the product).
CPU restatement of ext.sdfgen.sdf_from_points (reference source: ext/sdfgen/sdf_from_points.cu; kNN through its kd-tree,
ext/common/kdtree_cuda.cu -- here scipy's cKDTree, an exact kNN just like it):
  vote  (ComputeSDFKernel :83-140)  nearest neighbour p0 of the query x:  s = |n0.(x-p0)| if |x-p0| < stdv * ref_std[p0] else |x-p0|;
        sign = + iff MORE than k/2 of the k neighbours have n_k.(x-p_k) > 0;
  IMLS  (ComputeIMLSKernel :32-81)  sum_k w_k n_k.(x-p_k) / sum_k w_k,  w_k = exp(-(|x-p_k|^2 - min_j |x-p_j|^2) / stdv^2);
  ref_std (:176-184)  1, or with adaptive_knn = a: mean distance of a reference point to its a nearest reference points (itself included).
PINNED ON THE REFERENCE ITSELF: the reference's extension compiles from its own sources for gfx950 (oracle/build_ref.py ->
oracle/_ref/nksr_sdfgen.so) and runs on the GPU box; tests/test_gpu_sdfgen.py::test_sdf_from_points_matches_the_reference_binary
checks this restatement against it -- values and gradients to 1.2e-7 (vote) / 3.3e-6 (IMLS) on 6 600 queries per mode, no
sign flips (profiles/r04_parity_report.txt)."""
import numpy as np
from scipy.spatial import cKDTree


def sdf_from_points(queries, ref_xyz, ref_normal, nb_points, stdv, compute_grad=False, imls=False, adaptive_knn=0):
    q = np.asarray(queries, np.float64)
    ref = np.asarray(ref_xyz, np.float64)
    nrm = np.asarray(ref_normal, np.float64)
    tree = cKDTree(ref)
    k = int(nb_points)
    ref_std = np.ones(len(ref))
    if adaptive_knn > 0:
        dd, _ = tree.query(ref, k=int(adaptive_knn))
        ref_std = np.atleast_2d(dd.T).T.reshape(len(ref), -1).mean(1)
    dist, idx = tree.query(q, k=k)
    dist, idx = dist.reshape(len(q), k), idx.reshape(len(q), k)
    ray = q[:, None, :] - ref[idx]                        # [Q, k, 3]
    d = (nrm[idx] * ray).sum(-1)                          # n_k . (x - p_k)
    if imls:
        e = (ray ** 2).sum(-1) / (stdv * stdv)
        w = np.exp(-e + e.min(1, keepdims=True))
        sdf = (d * w).sum(1) / w.sum(1)
        grad = (nrm[idx] * w[..., None]).sum(1) / w.sum(1)[:, None]
    else:
        n0, r0, d0, l0 = nrm[idx[:, 0]], ray[:, 0], d[:, 0], dist[:, 0]
        near = l0 < stdv * ref_std[idx[:, 0]]
        mag = np.where(near, np.abs(d0), l0)
        g0 = np.where(near[:, None], np.where(d0[:, None] > 0, n0, -n0), r0 / np.maximum(l0, 1e-300)[:, None])
        sign = np.where((d > 0).sum(1) <= k // 2, -1.0, 1.0)
        sdf, grad = sign * mag, sign[:, None] * g0
    return (sdf.astype(np.float32), grad.astype(np.float32)) if compute_grad else (sdf.astype(np.float32),)
