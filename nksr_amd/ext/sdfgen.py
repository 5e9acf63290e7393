"""``sdf_from_points``: signed distance of query points to an oriented point cloud, from each query's k nearest reference
points (reference: ext/sdfgen/sdf_from_points.cu:142-235, bound in ext/sdfgen/bind.cpp:10-15; call sites models/loss.py:85
``sdf_from_points(query_pos, ref_xyz, ref_normal, 8, 0.02, False)[0]`` and dataset/av_gt_geometry.py:64-76
``nb_points=8, stdv=3.0, adaptive_knn=8``).

The reference builds a CUDA kd-tree (tinyflann), writes the k indices of every query and runs one of two kernels over them.
Here: the reference cloud is binned once into a uniform grid (Morton sort + cell hash, ``normals.PointGrid``), and ONE kernel per
call (csrc/knn.hip ``k_sdf_from_points``) finds every query's k-th neighbour distance by bisection over the cells around it and
evaluates the estimator while scanning the same cells -- no index lists.  Queries farther than ``max_ring`` cells from k
reference points are retried on a coarser grid (x4 per round), so every query gets an answer like with a kd-tree.
"""
import torch

from ..normals import PointGrid, choose_cell_size
from .._lib import call, ptr, stream


_MAX_ROUNDS = 16        # the cell size grows 4x per round: 4^16 cells of the first size span any finite cloud


def _grid_args(pg):
    h = pg.grid.hash
    return ptr(pg.start), ptr(pg.end), ptr(h.hkeys), ptr(h.hvals), h.cap, pg.cell, pg.inv_cell


def _mean_knn_distance(ref_xyz, k, cell):
    """Per reference point (original order): mean distance to its k nearest reference points, itself included."""
    n = ref_xyz.shape[0]
    out = torch.zeros(n, dtype=torch.float32, device=ref_xyz.device)
    todo = torch.arange(n, device=ref_xyz.device)
    for _ in range(_MAX_ROUNDS):
        if not todo.numel():
            break
        pg = PointGrid(ref_xyz, cell)
        std = torch.empty(n, dtype=torch.float32, device=ref_xyz.device)
        valid = torch.empty(n, dtype=torch.int32, device=ref_xyz.device)
        call('nksr_knn_mean_dist', ptr(pg.xyz), n, *_grid_args(pg), int(k), 4, ptr(std), ptr(valid), stream())
        back = torch.empty(n, dtype=torch.float32, device=ref_xyz.device)
        ok = torch.empty(n, dtype=torch.bool, device=ref_xyz.device)
        back[pg.perm] = std
        ok[pg.perm] = valid > 0
        sel = todo[ok[todo]]
        out[sel] = back[sel]
        todo = todo[~ok[todo]]
        cell *= 4.0
    if todo.numel():
        raise RuntimeError('sdf_from_points: %d reference points found no %d neighbours' % (todo.numel(), k))
    return out


def sdf_from_points(queries, ref_xyz, ref_normal, nb_points, stdv, compute_grad=False, imls=False, adaptive_knn=0):
    """-> [sdf [Q]] or [sdf [Q], grad [Q, 3]] (float32), the reference's return convention."""
    if not (queries.is_cuda and ref_xyz.is_cuda and ref_normal.is_cuda):
        raise RuntimeError('sdf_from_points: GPU tensors required (MI355X-only)')
    k = int(nb_points)
    n = ref_xyz.shape[0]
    if n < max(k, int(adaptive_knn), 1):
        raise RuntimeError('sdf_from_points: %d reference points for nb_points=%d' % (n, k))
    dev = queries.device
    if not (bool(torch.isfinite(queries).all()) and bool(torch.isfinite(ref_xyz).all()) and bool(torch.isfinite(ref_normal).all())):
        raise RuntimeError('sdf_from_points: non-finite input')
    q = queries.to(torch.float32).contiguous()
    ref = ref_xyz.to(torch.float32).contiguous()
    nrm = ref_normal.to(torch.float32).contiguous()
    cell = choose_cell_size(ref, max(k, int(adaptive_knn), 8))
    ref_std = _mean_knn_distance(ref, int(adaptive_knn), cell) if int(adaptive_knn) > 0 else None
    nq = q.shape[0]
    sdf = torch.zeros(nq, dtype=torch.float32, device=dev)
    grad = torch.zeros((nq, 3), dtype=torch.float32, device=dev) if compute_grad else None
    todo = torch.arange(nq, device=dev)
    for _ in range(_MAX_ROUNDS):
        if not todo.numel():
            break
        pg = PointGrid(ref, cell)
        ns = nrm[pg.perm].contiguous()
        stds = ref_std[pg.perm].contiguous() if ref_std is not None else None
        qs = q[todo].contiguous()
        m = qs.shape[0]
        s = torch.empty(m, dtype=torch.float32, device=dev)
        g = torch.empty((m, 3), dtype=torch.float32, device=dev) if compute_grad else None
        valid = torch.empty(m, dtype=torch.int32, device=dev)
        call('nksr_sdf_from_points', ptr(pg.xyz), ptr(ns), ptr(stds), *_grid_args(pg), ptr(qs), m, k, 4, float(stdv), int(bool(imls)),
             ptr(s), ptr(g), ptr(valid), stream())
        ok = valid > 0
        sdf[todo[ok]] = s[ok]
        if compute_grad:
            grad[todo[ok]] = g[ok]
        todo = todo[~ok]
        cell *= 4.0
    if todo.numel():
        raise RuntimeError('sdf_from_points: %d queries found no %d neighbours' % (todo.numel(), k))
    return [sdf, grad] if compute_grad else [sdf]
