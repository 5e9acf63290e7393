"""``sdf_from_points``: signed distance of query points to an oriented point cloud, from each query's k nearest reference
points (reference: ext/sdfgen/sdf_from_points.cu:142-235, bound in ext/sdfgen/bind.cpp:10-15; call sites models/loss.py:85
``sdf_from_points(query_pos, ref_xyz, ref_normal, 8, 0.02, False)[0]`` and dataset/av_gt_geometry.py:64-76
``nb_points=8, stdv=3.0, adaptive_knn=8``).

The reference builds a CUDA kd-tree (tinyflann), writes the k indices of every query and runs one of two kernels over them.
Here (nb_points <= 32): the reference cloud is binned once into a uniform grid (Morton sort + cell hash, ``normals.PointGrid``), an
octree is stacked on it (``normals.PointPyramid``: the same sorted points, cells x2 per level), and ONE kernel per call
(csrc/knn.hip ``k_sdf_pyramid``) takes every query to the scale at which the cloud is within one cell of it, keeps its k nearest
candidates sorted in registers while it descends the cells around it with box pruning, and evaluates the estimator from them -- no
index lists, no second pass.  What is left (queries farther from the cloud than 4 cells of the coarsest level; nb_points > 32,
which bisects for the k-th distance instead) goes through single grids 4x coarser per round, so every query gets an answer like
with a kd-tree.  Measurement knobs (tests/sdfgen_vs_ref.py --variants): ``NKSR_SDFGEN_SEARCH=rounds`` keeps everything on that path,
``NKSR_SDFGEN_LEAF`` / ``NKSR_SDFGEN_RINGS`` set the octree's scan threshold (48) and the rings searched per level (4).
"""
import os

import torch

from ..normals import PointGrid, PointPyramid, choose_cell_size
from .._lib import call, ptr, stream


_MAX_ROUNDS = 16        # the cell size grows 4x per round: 4^16 cells of the first size span any finite cloud
_PYRAMID_MAX_K = 32     # csrc/knn.hip keeps the candidates of k <= 32 in registers


def _use_pyramid(k):
    return 0 < k <= _PYRAMID_MAX_K and os.environ.get('NKSR_SDFGEN_SEARCH', 'pyramid') != 'rounds'


def _pyramid_knobs():
    return int(os.environ.get('NKSR_SDFGEN_LEAF', '0')), int(os.environ.get('NKSR_SDFGEN_RINGS', '4'))


def _grid_args(pg):
    h = pg.grid.hash
    return ptr(pg.start), ptr(pg.end), ptr(h.hkeys), ptr(h.hvals), h.cap, pg.cell, pg.inv_cell


def _mean_knn_distance(ref_xyz, k, cell, pyramid=None):
    """Per reference point (original order): mean distance to its k nearest reference points, itself included."""
    n = ref_xyz.shape[0]
    out = torch.zeros(n, dtype=torch.float32, device=ref_xyz.device)
    todo = torch.arange(n, device=ref_xyz.device)
    if _use_pyramid(k):
        pg = pyramid.pg
        leaf, rings = _pyramid_knobs()
        std = torch.empty(n, dtype=torch.float32, device=ref_xyz.device)
        valid = torch.empty(n, dtype=torch.int32, device=ref_xyz.device)
        call('nksr_knn_mean_dist_pyramid', pyramid.struct, n, int(k), rings, ptr(std), ptr(valid), stream())
        out[pg.perm] = std
        if bool((valid > 0).all()):
            return out
        ok = torch.empty(n, dtype=torch.bool, device=ref_xyz.device)
        ok[pg.perm] = valid > 0
        todo = todo[~ok]
        cell = pyramid.top_cell * 2.0
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
    if not bool(torch.isfinite(queries).all() & torch.isfinite(ref_xyz).all() & torch.isfinite(ref_normal).all()):      # (one readback)
        raise RuntimeError('sdf_from_points: non-finite input')
    q = queries.to(torch.float32).contiguous()
    ref = ref_xyz.to(torch.float32).contiguous()
    nrm = ref_normal.to(torch.float32).contiguous()
    cell = choose_cell_size(ref, max(k, int(adaptive_knn), 8))
    pyramid = None
    if _use_pyramid(k) or _use_pyramid(int(adaptive_knn)):
        leaf, rings = _pyramid_knobs()
        pyramid = PointPyramid(PointGrid(ref, cell), leaf=leaf)
    ref_std = _mean_knn_distance(ref, int(adaptive_knn), cell, pyramid) if int(adaptive_knn) > 0 else None
    nq = q.shape[0]
    sdf = torch.zeros(nq, dtype=torch.float32, device=dev)
    grad = torch.zeros((nq, 3), dtype=torch.float32, device=dev) if compute_grad else None
    todo = None
    if _use_pyramid(k):
        pg = pyramid.pg
        ns = nrm[pg.perm].contiguous()
        stds = ref_std[pg.perm].contiguous() if ref_std is not None else None
        valid = torch.empty(nq, dtype=torch.int32, device=dev)
        call('nksr_sdf_from_points_pyramid', pyramid.struct, ptr(ns), ptr(stds), ptr(q), nq, k, rings, float(stdv), int(bool(imls)),
             ptr(sdf), ptr(grad), ptr(valid), stream())
        if bool((valid > 0).all()):
            return [sdf, grad] if compute_grad else [sdf]
        todo = torch.nonzero(valid == 0).flatten()
        cell = pyramid.top_cell * 2.0
    if todo is None:
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
