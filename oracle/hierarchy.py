"""Oracle: sparse voxel hierarchy (restates nksr.SparseFeatureHierarchy).

Reference anchors: constructor + build_point_splatting models/nksr_net.py:57-62;
``grids[d].active_grid_coords()`` / ``grid_to_world`` / ``voxel_size``
models/loss.py:36,45-46; ``get_voxel_centers(d)`` models/nksr_net.py:100.
Canonical order of the voxels of a level = ascending Morton key (spec.morton_key).
"""
import numpy as np
from . import spec


class Level:
    def __init__(self, keys, level, voxel_size):
        self.level = level
        self.voxel_size = float(voxel_size) * (1 << level)
        self.keys = np.ascontiguousarray(keys, np.int64)  # sorted unique
        self.ijk = spec.morton_decode(self.keys, level)
        self.n = int(self.keys.shape[0])
        self.nbr = None

    def lookup(self, ijk):
        """Voxel index of integer coordinates (or -1)."""
        k = spec.morton_key(ijk, self.level)
        pos = np.searchsorted(self.keys, k)
        pos_c = np.minimum(pos, max(self.n - 1, 0))
        hit = (pos < self.n) & (self.keys[pos_c] == k) if self.n else np.zeros(k.shape, bool)
        return np.where(hit, pos_c, -1).astype(np.int32)

    def build_nbr(self):
        nb = np.empty((self.n, 27), np.int32)
        for s, o in enumerate(spec.NBR_OFFSETS):
            nb[:, s] = self.lookup(self.ijk + o)
        self.nbr = nb
        return nb

    def centers(self):
        return ((self.ijk.astype(np.float32) + np.float32(0.5)) * np.float32(self.voxel_size)).astype(np.float32)


class Hierarchy:
    def __init__(self, voxel_size, depth):
        self.voxel_size = float(voxel_size)
        self.depth = int(depth)
        self.levels = [None] * depth

    # -- builders ---------------------------------------------------------------------
    def build_point_splatting(self, xyz):
        """Activate, at every level, the 8 voxel centres nearest to each point."""
        H0, _ = spec.half_index(xyz, self.voxel_size)
        for d in range(self.depth):
            Hd = H0 >> d
            base = (Hd - 1) >> 1
            ijk = (base[:, None, :] + spec.CORNER_OFFSETS[None]).reshape(-1, 3)
            keys = np.unique(spec.morton_key(ijk, d))
            self.levels[d] = Level(keys, d, self.voxel_size)
        self.finalize()
        return self

    def build_adaptive_normal_variation(self, xyz, normal, tau=0.1, adaptive_depth=1):
        """Training-GT structure (models/nksr_net.py:175-179): levels below adaptive_depth are splatted
        only from points whose next-coarser cell has normal variation 1 - |mean normal| > tau."""
        H0, _ = spec.half_index(xyz, self.voxel_size)
        alive = np.ones(xyz.shape[0], bool)
        for d in range(self.depth - 1, -1, -1):
            if d < adaptive_depth and d + 1 < self.depth:
                cell = spec.morton_key((H0 >> (d + 1)) >> 1, d + 1)
                _, inv, cnt = np.unique(cell, return_inverse=True, return_counts=True)
                acc = np.zeros((len(cnt), 3), np.float64)
                np.add.at(acc, inv, normal.astype(np.float64))
                variation = 1.0 - np.linalg.norm(acc, axis=1) / cnt
                alive = alive & (variation > tau)[inv]
            base = ((H0[alive] >> d) - 1) >> 1
            ijk = (base[:, None, :] + spec.CORNER_OFFSETS[None]).reshape(-1, 3)
            self.levels[d] = Level(np.unique(spec.morton_key(ijk, d)), d, self.voxel_size)
        self.finalize()
        return self

    def build_point_neighborhood(self, xyz):
        """Activate, at every level, the cell containing each point and its 26
        neighbours (the analytic structure rule of the decoder hierarchy,
        DESIGN.md section 2.2)."""
        H0, _ = spec.half_index(xyz, self.voxel_size)
        for d in range(self.depth):
            Id = (H0 >> d) >> 1
            ijk = (Id[:, None, :] + spec.NBR_OFFSETS[None]).reshape(-1, 3)
            keys = np.unique(spec.morton_key(ijk, d))
            self.levels[d] = Level(keys, d, self.voxel_size)
        self.finalize()
        return self

    def build_from_keys(self, keys_per_level):
        for d in range(self.depth):
            self.levels[d] = Level(np.unique(np.asarray(keys_per_level[d], np.int64)), d, self.voxel_size)
        self.finalize()
        return self

    def finalize(self):
        off = 0
        self.offsets = []
        for d, L in enumerate(self.levels):
            L.build_nbr()
            self.offsets.append(off)
            off += L.n
        self.num_unknowns = off
        for d, L in enumerate(self.levels):
            if d + 1 < self.depth:
                L.parent = self.levels[d + 1].lookup(L.ijk >> 1)
            else:
                L.parent = np.full(L.n, -1, np.int32)

    # -- queries ----------------------------------------------------------------------
    def evaluate_voxel_status(self, query_ijk, depth):
        """Class of every voxel of a level-``depth`` query grid w.r.t. this hierarchy (models/loss.py:155: ground truth of the
        structure cross-entropy): 0 = not a voxel here, 1 = voxel without children one level finer, 2 = voxel with children."""
        lv = self.levels[depth]
        status = np.zeros(query_ijk.shape[0], np.int64)
        if lv is None or lv.n == 0 or query_ijk.shape[0] == 0:
            return status
        exist = lv.lookup(query_ijk) >= 0
        status[exist] = 1
        if depth > 0 and self.levels[depth - 1] is not None and self.levels[depth - 1].n:
            parents = set(map(tuple, (self.levels[depth - 1].ijk >> 1)))      # floor(ijk / 2): the level-depth voxel a finer voxel sits in
            has_child = np.fromiter((tuple(q) in parents for q in query_ijk), bool, query_ijk.shape[0])
            status[exist & has_child] = 2
        return status

    def get_voxel_centers(self, d):
        return self.levels[d].centers()

    def site_cells(self, xyz):
        """Per level: containing-cell voxel index (or -1), local coordinate u in [0,1)
        and the half bit hb (1 when x lies in the upper half of its cell)."""
        H0, p = spec.half_index(xyz, self.voxel_size)
        out = []
        for d in range(self.depth):
            Hd = H0 >> d
            Id = Hd >> 1
            pd = p * np.float32(2.0 ** (-d))
            u = (pd - Id.astype(np.float32)).astype(np.float32)
            hb = (Hd & 1).astype(np.int32)
            cell = self.levels[d].lookup(Id)
            out.append((cell, u, hb))
        return out
