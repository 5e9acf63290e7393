"""Sparse voxel hierarchy -- host-side mirror of ``nksr.svh.SparseFeatureHierarchy``.

Reference interface (call sites; the implementation ships in the absent wheel):
  * ``SparseFeatureHierarchy(voxel_size=, depth=, device=)`` + ``build_point_splatting(xyz)``
    -- models/nksr_net.py:57-62
  * ``grids[d]`` (``None`` when a level is empty) with ``active_grid_coords()``,
    ``grid_to_world(ijk.float())``, ``voxel_size`` -- models/loss.py:33-46
  * ``get_voxel_centers(d)`` -- models/nksr_net.py:100; ``depth`` / ``device`` -- models/loss.py:33,39
  * ``evaluate_voxel_status(grid, d)`` -> {0 non-exist, 1 exist-stop, 2 exist-continue}
    -- models/loss.py:155-160
Level d has voxel width voxel_size * 2**d; the canonical order of a level's voxels is the
ascending Morton key order produced by the device radix sort, so topology is index-exact and
independent of atomics.  All heavy lifting happens in nksr_amd/csrc/hierarchy.hip.
"""
import enum
import os

import numpy as np
import torch

from . import _lib, ops
from ._lib import call, ptr, stream


_DEDUP_MIN = 1 << 19     # keys: below this a key stream is sorted as it is (the dedup pass costs a host sync for its count)


class VoxelStatus(enum.IntEnum):
    VS_NON_EXIST = 0
    VS_EXIST_STOP = 1
    VS_EXIST_CONTINUE = 2


def inv_w0_f32(voxel_size):
    return float(np.float32(1.0 / float(voxel_size)))


class SparseGrid:
    """One level: sorted Morton keys, ijk, hash table and the 27-neighbour table."""

    def __init__(self, keys, level, base_voxel_size, coarse=None):
        """``coarse``: the next-coarser level of the same hierarchy, already built -- the neighbour table is then derived from ITS
        table and the children of its voxels (csrc/hierarchy.hip: k_build_nbr_parent) instead of 27 hash probes per voxel; the
        result is the same table (a level whose voxels lack parents falls back to the hash by itself, on the device)."""
        self.level = level
        self.voxel_size = float(base_voxel_size) * (1 << level)
        self.keys = keys
        self.device = keys.device
        self.num_voxels = n = int(keys.numel())
        self.ijk = torch.empty((n, 3), dtype=torch.int32, device=self.device)
        call('nksr_decode_keys', ptr(keys), n, level, ptr(self.ijk), stream())
        self.hash = ops.HashTable(keys)
        self.nbr = torch.empty((n, 27), dtype=torch.int32, device=self.device)
        if coarse is not None and coarse.num_voxels > 0 and n > 0 and coarse.level == level + 1 and os.environ.get('NKSR_NBR_FROM_PARENT', '1') != '0':
            parent = coarse.hash.query((keys >> 3).contiguous())
            work = torch.zeros(2 * coarse.num_voxels + 1, dtype=torch.int32, device=self.device)
            call('nksr_build_nbr_from_parent', ptr(self.ijk), ptr(keys), n, level, ptr(self.hash.hkeys), ptr(self.hash.hvals), self.hash.cap,
                 ptr(parent), ptr(coarse.nbr), coarse.num_voxels, ptr(work), ptr(self.nbr), stream())
        else:
            call('nksr_build_nbr', ptr(self.ijk), n, level, ptr(self.hash.hkeys), ptr(self.hash.hvals), self.hash.cap,
                 ptr(self.nbr), stream())

    def active_grid_coords(self):
        return self.ijk

    def grid_to_world(self, ijk):
        return (ijk.to(torch.float32) + 0.5) * self.voxel_size

    def world_to_grid(self, xyz):
        return xyz / self.voxel_size - 0.5

    def ijk_to_index(self, ijk):
        """Canonical voxel index of integer coordinates (-1 where inactive)."""
        ijk = ijk.to(torch.int32).contiguous()
        keys = torch.empty(ijk.shape[0], dtype=torch.int64, device=self.device)
        call('nksr_encode_keys', ptr(ijk), ijk.shape[0], self.level, ptr(keys), stream())
        return self.hash.query(keys)

    def to(self, device):
        g = object.__new__(SparseGrid)
        g.__dict__.update(self.__dict__)
        for k in ('keys', 'ijk', 'nbr'):
            setattr(g, k, getattr(self, k).to(device))
        g.hash = object.__new__(ops.HashTable)
        g.hash.cap = self.hash.cap
        g.hash.hkeys, g.hash.hvals = self.hash.hkeys.to(device), self.hash.hvals.to(device)
        g.device = torch.device(device)
        return g


class SparseFeatureHierarchy:
    def __init__(self, voxel_size, depth, device):
        self.device = _lib.require_gpu(device)
        if not (1 <= depth <= _lib.MAX_DEPTH):
            raise RuntimeError('depth must be in [1, %d]' % _lib.MAX_DEPTH)
        self.voxel_size = float(voxel_size)
        self.depth = int(depth)
        self._levels = [None] * self.depth
        self.inv_w0 = inv_w0_f32(voxel_size)

    # ---- builders ---------------------------------------------------------------------------
    def _check_xyz(self, xyz):
        if xyz.dtype != torch.float32 or xyz.dim() != 2 or xyz.shape[1] != 3:
            raise RuntimeError('xyz must be a float32 [N,3] tensor')
        _lib.require_gpu(xyz.device)
        if xyz.shape[0] and getattr(ops._tls, 'hint', None) is None:        # (under a key_hint the caller has checked the cloud's box already)
            amax = float(xyz.abs().max())
            if not (amax * self.inv_w0 < (1 << 20) - 8):     # also catches NaN / inf
                raise RuntimeError('coordinates out of range: |x| / voxel_size must stay below 2^20 (got %g); '
                                   'recentre the cloud or use a larger voxel_size' % (amax * self.inv_w0))
        return xyz.contiguous()

    def _build_from_points(self, xyz, mode):
        xyz = self._check_xyz(xyz)
        n = xyz.shape[0]
        per = 8 if mode == 0 else 27
        for d in range(self.depth - 1, -1, -1):          # coarse -> fine: a level's neighbour table comes from the level above it
            raw = torch.empty(n * per, dtype=torch.int64, device=self.device)
            call('nksr_splat_keys', ptr(xyz), n, self.inv_w0, d, mode, ptr(raw), stream())
            keys = ops.sort_unique(raw, level=d)
            self._levels[d] = SparseGrid(keys, d, self.voxel_size, coarse=self._coarse(d))
        return self

    def _coarse(self, d):
        return self._levels[d + 1] if d + 1 < self.depth else None

    def build_point_splatting(self, xyz):
        """Activate the 8 voxel centres nearest to every point at every level
        (reference: models/nksr_net.py:62)."""
        return self._build_from_points(xyz, 0)

    def build_point_neighborhood(self, xyz):
        """Activate the containing cell and its 26 neighbours at every level: the analytic
        structure rule of the decoder hierarchy."""
        return self._build_from_points(xyz, 1)

    @staticmethod
    def cells_with_points(point_keys_sorted, depth):
        """Per level: the unique cells that contain a point, from the SORTED level-0 point keys.  Level d + 1
        is derived from the (much shorter) level-d list: the parent of a cell is ``key >> 3``."""
        cells = [ops.unique_sorted(point_keys_sorted)]
        for d in range(1, depth):
            cells.append(ops.unique_sorted((cells[-1] >> 3).contiguous()))
        return cells

    def _footprint(self, cells, level, mode):
        per = 8 if mode == 0 else 27
        raw = torch.empty(cells.numel() * per, dtype=torch.int64, device=self.device)
        if raw.numel() >= _DEDUP_MIN:
            return ops.sort_unique(self._dedup(None, cells, cells.numel(), level, mode, raw), level=level + (1 if mode == 0 else 0))
        call('nksr_cell_footprint_keys', ptr(cells), cells.numel(), level, mode, ptr(raw), stream())
        return ops.sort_unique(raw, level=level + (1 if mode == 0 else 0))

    def _dedup(self, xyz, cells, n, level, mode, raw):
        """Large key streams: duplicates of neighbouring (Morton-ordered) elements are dropped in LDS before the sort
        (nksr_footprint_keys_dedup): the level-0 streams of a 1 M-point cloud shrink from 8-11 M keys to 1-2 M."""
        cnt = torch.empty(1, dtype=torch.int64, device=self.device)
        call('nksr_footprint_keys_dedup', ptr(xyz) if xyz is not None else None, ptr(cells) if cells is not None else None, n,
             self.inv_w0, level, mode, ptr(raw), ptr(cnt), stream())
        return raw[:int(cnt.item())]

    def build_point_splatting_sorted(self, xyz_sorted, point_keys_sorted, cells=None):
        """Same result as build_point_splatting, 3-8x fewer keys to sort: level 0 from the points,
        level d >= 1 from the unique level-(d-1) cells (their index is the level-d half index).
        ``cells``: result of cells_with_points (shared with build_point_neighborhood_sorted)."""
        xyz_sorted = self._check_xyz(xyz_sorted)
        n = xyz_sorted.shape[0]
        raw = torch.empty(n * 8, dtype=torch.int64, device=self.device)
        if raw.numel() >= _DEDUP_MIN:
            raw = self._dedup(xyz_sorted, None, n, 0, 0, raw)
        else:
            call('nksr_splat_keys', ptr(xyz_sorted), n, self.inv_w0, 0, 0, ptr(raw), stream())
        keys0 = ops.sort_unique(raw, level=0)
        if cells is None and self.depth > 1:
            cells = self.cells_with_points(point_keys_sorted, self.depth - 1)
        for d in range(self.depth - 1, 0, -1):
            self._levels[d] = SparseGrid(self._footprint(cells[d - 1], d - 1, 0), d, self.voxel_size, coarse=self._coarse(d))
        self._levels[0] = SparseGrid(keys0, 0, self.voxel_size, coarse=self._coarse(0))
        return self

    def build_point_neighborhood_sorted(self, point_keys_sorted, cells=None):
        """Same result as build_point_neighborhood from the unique cells that hold points."""
        if cells is None:
            cells = self.cells_with_points(point_keys_sorted, self.depth)
        for d in range(self.depth - 1, -1, -1):
            self._levels[d] = SparseGrid(self._footprint(cells[d], d, 1), d, self.voxel_size, coarse=self._coarse(d))
        return self

    def build_adaptive_normal_variation(self, xyz, normal, tau=0.1, adaptive_depth=1):
        """Ground-truth structure for the structure loss (models/nksr_net.py:175-179,
        configs/default/train.yaml:45-47).  Levels d >= adaptive_depth: plain point splatting.  A finer
        level d < adaptive_depth is splatted only from the points whose level-(d+1) cell is *varied*:
        1 - |mean unit normal of the cell's points| > tau (and whose coarser cells were varied too),
        so flat regions stop at a coarse level.  (Spec: DESIGN.md section 2.2; implementation absent
        from the reference tree.)"""
        from .nn.network import sort_cloud
        xyz = self._check_xyz(xyz)
        normal = normal.to(torch.float32).contiguous()
        ks, xs, ns = sort_cloud(xyz, normal, self.inv_w0)
        n = xs.shape[0]
        alive = torch.ones(n, dtype=torch.bool, device=self.device)
        for d in range(self.depth - 1, -1, -1):
            if d < adaptive_depth and d + 1 < self.depth:
                cell = ks >> (3 * (d + 1))
                _, inv, cnt = torch.unique_consecutive(cell, return_inverse=True, return_counts=True)
                acc = torch.zeros((cnt.numel(), 3), dtype=torch.float64, device=self.device)
                acc.index_add_(0, inv, ns.double())
                variation = 1.0 - acc.norm(dim=1) / cnt.double()
                alive = alive & (variation > float(tau))[inv]
            pts = xs if bool(alive.all()) else xs[alive].contiguous()
            raw = torch.empty(pts.shape[0] * 8, dtype=torch.int64, device=self.device)
            call('nksr_splat_keys', ptr(pts), pts.shape[0], self.inv_w0, d, 0, ptr(raw), stream())
            self._levels[d] = SparseGrid(ops.sort_unique(raw), d, self.voxel_size, coarse=self._coarse(d))
        return self

    def build_from_keys(self, keys_per_level, sorted_unique=False):
        """``sorted_unique``: the keys are already in canonical order (a packed field's payload)."""
        for d in range(self.depth - 1, -1, -1):
            k = keys_per_level[d]
            if k is None:
                k = torch.empty(0, dtype=torch.int64, device=self.device)
            k = k.to(self.device).contiguous()
            self._levels[d] = SparseGrid(k if sorted_unique else ops.sort_unique(k), d, self.voxel_size, coarse=self._coarse(d))
        return self

    def build_from_grid_coords(self, depth, ijk):
        ijk = ijk.to(device=self.device, dtype=torch.int32).contiguous()
        keys = torch.empty(ijk.shape[0], dtype=torch.int64, device=self.device)
        call('nksr_encode_keys', ptr(ijk), ijk.shape[0], depth, ptr(keys), stream())
        self._levels[depth] = SparseGrid(ops.sort_unique(keys), depth, self.voxel_size)
        return self

    # ---- queries ------------------------------------------------------------------------------
    @property
    def grids(self):
        """Per level: the grid, or ``None`` when the level holds no voxel (reference
        convention, models/nksr_net.py:80, models/loss.py:34)."""
        return [g if (g is not None and g.num_voxels > 0) else None for g in self._levels]

    def level(self, d):
        """Level d as a (possibly empty) SparseGrid."""
        if self._levels[d] is None:
            self._levels[d] = SparseGrid(torch.empty(0, dtype=torch.int64, device=self.device), d, self.voxel_size)
        return self._levels[d]

    def num_voxels(self, d):
        return 0 if self._levels[d] is None else self._levels[d].num_voxels

    @property
    def offsets(self):
        off, o = [], 0
        for d in range(self.depth):
            off.append(o)
            o += self.num_voxels(d)
        return off

    @property
    def num_unknowns(self):
        return sum(self.num_voxels(d) for d in range(self.depth))

    def get_voxel_centers(self, d):
        g = self.grids[d]
        if g is None:
            return torch.zeros((0, 3), dtype=torch.float32, device=self.device)
        return g.grid_to_world(g.ijk)

    def evaluate_voxel_status(self, grid, depth):
        """Class id of every voxel of ``grid`` (a level-``depth`` grid) w.r.t. this hierarchy."""
        n = grid.num_voxels
        status = torch.zeros(n, dtype=torch.long, device=self.device)
        mine = self.grids[depth]
        if mine is None or n == 0:
            return status
        exist = mine.hash.query(grid.keys) >= 0
        status[exist] = int(VoxelStatus.VS_EXIST_STOP)
        if depth > 0 and self.grids[depth - 1] is not None:
            child_parent = ops.sort_unique(self.grids[depth - 1].keys >> 3)
            has_child = ops.sorted_lookup(child_parent, grid.keys.contiguous()) >= 0
            status[exist & has_child] = int(VoxelStatus.VS_EXIST_CONTINUE)
        return status

    def get_visualization(self):
        """Wireframe-free stand-in for the reference's pycg visualisation: list of
        (voxel centres, voxel size) per level."""
        return [(self.get_voxel_centers(d).cpu().numpy(), self.voxel_size * (1 << d)) for d in range(self.depth)]

    def to_(self, device):
        device = torch.device(device)
        self._levels = [None if g is None else g.to(device) for g in self._levels]
        self.device = device
        return self

    def __repr__(self):
        return 'SparseFeatureHierarchy(voxel_size=%g, depth=%d, voxels=%s)' % (
            self.voxel_size, self.depth, [self.num_voxels(d) for d in range(self.depth)])
