"""kNN-PCA normal estimation + sensor orientation on the GPU (csrc/knn.hip).

Mirrors the recipe the reference ships in source form (examples/recons_waymo_cpu.py:21-41, the
stand-in for ``nksr.get_estimate_normal_preprocess_fn(64, 85.0)``, examples/recons_waymo.py:36):
  1. unoriented normals = smallest PCA eigenvector of the k nearest neighbours (k includes the
     point itself)
  2. flip so that the normal faces the sensor:  (sensor - xyz) . n >= 0
  3. drop grazing points: keep |cos| > cos(deg)
"""
import math

import torch

from . import ops
from ._lib import call, ptr, stream
from .svh import SparseGrid, inv_w0_f32


class PointGrid:
    """Uniform grid over a cloud: Morton-sorted points + per-cell ranges + cell hash."""

    def __init__(self, xyz, cell):
        xyz = xyz.to(torch.float32).contiguous()
        n = xyz.shape[0]
        dev = xyz.device
        self.cell = float(cell)
        self.inv_cell = inv_w0_f32(cell)
        keys = torch.empty(n, dtype=torch.int64, device=dev)
        call('nksr_point_keys', ptr(xyz), n, self.inv_cell, ptr(keys), stream())
        ks, perm = ops.sort_pairs(keys, torch.arange(n, dtype=torch.int32, device=dev))
        self.perm = perm.long()
        self.xyz = xyz[self.perm].contiguous()
        self.keys_sorted = ks
        self.grid = SparseGrid(ops.unique_sorted(ks), 0, cell)
        self.start = torch.empty(self.grid.num_voxels, dtype=torch.int32, device=dev)
        self.end = torch.empty(self.grid.num_voxels, dtype=torch.int32, device=dev)
        call('nksr_site_ranges', ptr(ks), n, ptr(self.grid.keys), self.grid.num_voxels, 0, ptr(self.start), ptr(self.end), stream())

    def nearest(self, query, max_ring=8):
        """Index (into the ORIGINAL cloud order) of the nearest point of every query."""
        q = query.to(torch.float32).contiguous()
        idx = torch.empty(q.shape[0], dtype=torch.int32, device=q.device)
        h = self.grid.hash
        call('nksr_nearest_index', ptr(self.xyz), ptr(self.start), ptr(self.end), ptr(h.hkeys), ptr(h.hvals), h.cap, self.cell,
             self.inv_cell, ptr(q), q.shape[0], int(max_ring), ptr(idx), stream())
        ok = idx >= 0
        out = torch.full_like(idx, -1, dtype=torch.int64)
        out[ok] = self.perm[idx[ok].long()]
        return out


class PointPyramid:
    """Octree over a ``PointGrid``: level l = cells of size cell * 2^l (keys = the grid's keys >> 3 l), each with its point range, the
    range of its children on the level below and their octant mask, and a key hash (csrc/knn.hip ``KnnPyramid``).  Built upward from
    the grid until a level has <= ``top_cells`` cells (one kernel + one hash per level; the level sizes come back to the host).
    Eight, not one: keys are biased coordinates, so the cells either side of a coordinate plane through the origin never merge."""

    def __init__(self, pg, leaf=0, top_cells=8):
        from ._lib import KNN_LEVELS, KnnPyramidT
        dev = pg.xyz.device
        self.pg = pg
        keys, start, end = pg.grid.keys, pg.start, pg.end
        self.keep = [keys, start, end]
        t = KnnPyramidT()
        t.xyz_sorted = ptr(pg.xyz)
        t.cell, t.inv_cell, t.leaf = pg.cell, pg.inv_cell, int(leaf)
        h = pg.grid.hash
        lvl = 0
        while True:
            t.start[lvl], t.end[lvl], t.hkeys[lvl], t.hvals[lvl], t.hcap[lvl] = ptr(start), ptr(end), ptr(h.hkeys), ptr(h.hvals), h.cap
            lvl += 1
            nc = keys.numel()
            if lvl == KNN_LEVELS or nc <= top_cells:
                break
            up = ops.unique_sorted(keys >> 3)
            n = up.numel()
            child = torch.empty(n + 1, dtype=torch.int32, device=dev)
            cmask = torch.empty(n, dtype=torch.uint8, device=dev)
            s2 = torch.empty(n, dtype=torch.int32, device=dev)
            e2 = torch.empty(n, dtype=torch.int32, device=dev)
            call('nksr_knn_pyramid_level', ptr(keys), nc, ptr(start), ptr(end), ptr(up), n, ptr(child), ptr(cmask), ptr(s2), ptr(e2), stream())
            h = ops.HashTable(up)
            t.child[lvl], t.cmask[lvl] = ptr(child), ptr(cmask)
            self.keep += [up, child, cmask, s2, e2, h]
            keys, start, end = up, s2, e2
        t.levels = lvl
        self.levels = lvl
        self.top_cell = pg.cell * (1 << (lvl - 1))
        self.struct = t


def choose_cell_size(xyz, k):
    """Cell size such that a ball of one cell radius holds ~2k surface samples: density from the
    occupied-voxel count at one probe resolution (points on a surface: count ~ area / cell^2)."""
    from .density import bbox_center, occupied_voxels
    n = xyz.shape[0]
    lo, hi, center = bbox_center(xyz)
    ext = float((hi - lo).max())
    # the probe voxels must hold several samples each or their count saturates at n and the density comes out as 1 / probe^2 whatever
    # the cloud (4 000 points at ext / 256: one point per cell, every kNN query ran to its outermost ring): a surface of area ~ ext^2
    # sampled n times has ~4 samples per voxel of size 4 ext / sqrt(n); from 1 M points on that is finer than ext / 256 and nothing changes
    probe = max(ext / 256.0, 4.0 * ext / math.sqrt(max(n, 1)), 1e-6)
    occ = max(occupied_voxels((xyz - lo[None]).contiguous(), probe), 1)
    area = occ * probe * probe                      # ~ surface area
    rho = n / max(area, 1e-20)
    return max(math.sqrt(2.0 * k / (math.pi * rho)), probe / 8)


class TooFewPoints(RuntimeError):
    """Fewer points than the neighbourhood size: chunk mode treats such a chunk as empty (nksr_amd/chunking.py)."""


def knn_pca(xyz, knn):
    """Unoriented kNN-PCA normals: (PointGrid, normal [n,3], r2 [n], valid [n]) in the grid's Morton order (``pg.perm`` maps back).
    ``r2`` is the squared distance of the k-th nearest neighbour (the point itself included): the neighbour SET of point i is
    {j : |x_i - x_j|^2 <= r2_i} -- the kernel keeps no index lists, this is what the parity test compares with an exact kd-tree."""
    n = xyz.shape[0]
    pg = PointGrid(xyz, choose_cell_size(xyz, knn))
    nrm = torch.empty((n, 3), dtype=torch.float32, device=xyz.device)
    r2 = torch.empty(n, dtype=torch.float32, device=xyz.device)
    valid = torch.empty(n, dtype=torch.int32, device=xyz.device)
    todo = torch.empty(n, dtype=torch.int32, device=xyz.device)       # one wavefront per query; the queries it hands back (too many candidates)
    h = pg.grid.hash
    call('nksr_knn_pca_normals', ptr(pg.xyz), n, ptr(pg.start), ptr(pg.end), ptr(h.hkeys), ptr(h.hvals), h.cap, pg.cell,
         pg.inv_cell, int(knn), 6, ptr(nrm), ptr(r2), ptr(valid), ptr(todo), stream())
    return pg, nrm, r2, valid


def estimate_normals_knn(xyz, normal, sensor, knn=64, deg=85.0):
    """(xyz, normal=None, sensor) -> (xyz', normal', None), the preprocess_fn contract of
    examples/recons_waymo_cpu.py:21-41."""
    if normal is not None:
        raise RuntimeError('normal already exists')
    if sensor is None:
        raise RuntimeError('please provide sensor positions for consistent orientations')
    n = xyz.shape[0]
    if n < knn:
        raise TooFewPoints('need at least knn=%d points' % knn)
    pg, nrm, r2, valid = knn_pca(xyz, knn)
    xs = pg.xyz
    ss = sensor.to(torch.float32)[pg.perm]
    view = ss - xs
    view = view / (torch.linalg.norm(view, dim=-1, keepdim=True) + 1e-6)
    cos = (view * nrm).sum(1)
    nrm = torch.where((cos < 0)[:, None], -nrm, nrm)
    keep = (cos.abs() > math.cos(math.radians(deg))) & (valid > 0)
    # return in the original point order (stable w.r.t. the input, like the CPU recipe)
    order = torch.argsort(pg.perm[keep])
    return xs[keep][order].contiguous(), nrm[keep][order].contiguous(), None
