"""Oracle: spatial chunking + partition-of-unity blend (TEST INFRASTRUCTURE ONLY).

Restates ``reconstruct(xyz, normal, detail_level=None, chunk_size=...)`` (reference call sites
examples/recons_by_chunk.py:26-30, recons_waymo.py:30-37 with ``# chunk_size=51.2``; "Tuning detail_level /
voxel_size is not supported if chunk_size is provided" NKSR-USAGE.md:137).  The reference implementation is
in the absent wheel; the specification restated here is SURVEY.md App. B7 / DESIGN.md section 5:
  * bounding box of the cloud cut into a grid of ``chunk_size`` cubes (origin = bbox minimum)
  * ov = max(overlap_ratio * chunk_size, 1.6 * coarsest voxel);  chunk c solves the points inside
    core_c +- 2 ov (along the split axes only), each with the full single-field pipeline (oracle.pipeline)
  * f(x) = sum_c w_c(x) f_c(x) / sum_c w_c(x),  w_c = prod over split axes of
    clamp((x - (lo - ov)) / 2ov, 0, 1) * clamp(((hi + ov) - x) / 2ov, 0, 1);  fp32, chunks in ascending id
  * meshing on the union of the chunks' finest levels (one global voxel lattice)
  * mask: plain LayerField of the union grid, or -- with UDF masks -- the OR of the chunk masks over the
    blend support.
  * the EXPLODED FRAME: chunk c is solved on x' = fl32(x + T_c), T_c = a whole number of coarsest
    voxels that moves the chunk's data corner into the chunk's own aligned cube of the lattice (the product solves all
    chunks of a rank as ONE block-diagonal system there); its field is evaluated at fl32(x + T_c), its voxels return to
    the global lattice by subtracting the integer translation.  Restated here chunk by chunk (the oracle has no batch).
numpy only; independent of nksr_amd/chunking.py (the product runs this on the GPU).
"""
import math

import numpy as np

from . import hierarchy, meshing, pipeline


def chunk_grid(lo, hi, chunk_size):
    return [max(1, int(math.ceil((hi[a] - lo[a]) / chunk_size - 1e-9))) for a in range(3)]


def _f32(v):
    return np.float32(v)


def chunk_ids(xyz, lo, chunk_size, grid):
    """Linear id of the core containing every point (fp32 arithmetic: (x - lo) * (1 / chunk_size))."""
    inv = _f32(1.0) / _f32(chunk_size)
    cid = np.zeros(xyz.shape[0], np.int64)
    for a in range(3):
        if grid[a] > 1:
            ia = np.clip(np.floor((xyz[:, a] - _f32(lo[a])) * inv).astype(np.int64), 0, grid[a] - 1)
        else:
            ia = 0
        cid = cid * grid[a] + ia
    return cid


SLOT_GAP = 6          # empty coarsest voxels between the data of two slots


class Frame:
    """Slots of the exploded frame: chunk (cx, cy, cz) owns the cube [s 2^S, (s + 1) 2^S)^3 of the finest lattice,
    s = (cx, cy, cz) - grid // 2, data corner at slot origin + 4 coarsest voxels;  2^S >= that + data extent + 2^depth alignment slack
    + SLOT_GAP / 2 coarsest voxels."""

    def __init__(self, voxel_size, depth, lo, grid, chunk_size, band):
        self.w0, self.depth, self.lo, self.grid, self.cs, self.band = float(voxel_size), int(depth), list(lo), list(grid), float(chunk_size), float(band)
        self.align = 1 << depth
        self.low = 2 * self.align        # the data starts 4 coarsest voxels inside its slot: every voxel of the chunk lies in the slot
        need = self.low + int(math.ceil((chunk_size + 2 * band) / voxel_size)) + 2 + self.align + ((SLOT_GAP // 2) << (depth - 1))
        self.S = depth
        while (1 << self.S) < need:
            self.S += 1

    def c3(self, c):
        g = self.grid
        return (c // (g[1] * g[2]), (c // g[2]) % g[1], c % g[2])

    def shift_cells(self, c):
        out = []
        for a, ca in enumerate(self.c3(c)):
            data_lo = self.lo[a] + ca * self.cs - self.band if self.grid[a] > 1 else self.lo[a]
            corner = int(math.floor(math.floor(data_lo / self.w0) / self.align)) * self.align
            out.append(((ca - self.grid[a] // 2) << self.S) + self.low - corner)
        return np.asarray(out, np.int64)

    def shift(self, c):
        return np.asarray([np.float32(float(t) * self.w0) for t in self.shift_cells(c)], np.float32)


class ChunkedField:
    def __init__(self, fields, cores, ov, lo, chunk_size, grid, voxel_size, frame, adaptive_depth=1):
        self.fields, self.cores, self.ov, self.lo, self.chunk_size, self.grid = fields, cores, float(ov), lo, float(chunk_size), grid
        self.voxel_size, self.frame = float(voxel_size), frame
        from . import spec
        # union of the chunks' voxels, back on the global lattice (integer translation: T_c is a whole number of voxels at every
        # level): the finest level and the coarser ones below adaptive_depth (LayerField(dec_svh, adaptive_depth), models/nksr_net.py:132)
        self.adaptive_depth = max(1, int(adaptive_depth))
        keys = [[] for _ in range(self.adaptive_depth)]
        for c, f in fields.items():
            for d in range(self.adaptive_depth):
                ld = f['hier'].levels[d]
                keys[d].append(spec.morton_key(ld.ijk.astype(np.int64) - (frame.shift_cells(c) >> d)[None], d))
        self.union = hierarchy.Hierarchy(voxel_size, self.adaptive_depth).build_from_keys(
            [np.concatenate(k) if k else np.zeros(0, np.int64) for k in keys])

    def weight(self, c, xyz):
        lo, hi = self.cores[c]
        inv = _f32(1.0) / _f32(2 * self.ov)
        w = np.ones(xyz.shape[0], np.float32)
        for a in range(3):
            if self.grid[a] > 1:
                x = xyz[:, a].astype(np.float32)
                up = np.clip((x - _f32(lo[a] - self.ov)) * inv, _f32(0), _f32(1))
                dn = np.clip((_f32(hi[a] + self.ov) - x) * inv, _f32(0), _f32(1))
                w = (w * up) * dn
        return w.astype(np.float32)

    def evaluate(self, xyz, grad=False):
        xyz = np.asarray(xyz, np.float32)
        n = xyz.shape[0]
        num = np.zeros(n, np.float32)
        den = np.zeros(n, np.float32)
        gnum = np.zeros((n, 3), np.float32) if grad else None
        for c in sorted(self.fields):
            w = self.weight(c, xyz)
            sel = np.nonzero(w > 0)[0]
            if len(sel) == 0:
                continue
            f, g = pipeline.evaluate(self.fields[c], (xyz[sel] + self.frame.shift(c)[None]).astype(np.float32), grad)
            num[sel] = (num[sel] + f * w[sel]).astype(np.float32)
            den[sel] = (den[sel] + w[sel]).astype(np.float32)
            if grad:
                gnum[sel] = (gnum[sel] + g * w[sel, None]).astype(np.float32)
        den = np.maximum(den, np.float32(1e-20))
        return (num / den).astype(np.float32), ((gnum / den[:, None]).astype(np.float32) if grad else None)

    def mask(self, xyz):
        from . import network as onet
        if not any(f.get('udf_feats') is not None for f in self.fields.values()):
            return None
        keep = np.zeros(xyz.shape[0], bool)
        for c in sorted(self.fields):
            f = self.fields[c]
            if f.get('udf_feats') is None:
                continue
            sel = np.nonzero(self.weight(c, xyz) > 0)[0]
            if len(sel):
                d = onet.udf_decode(f['hier'], f['udf_feats'], (xyz[sel] + self.frame.shift(c)[None]).astype(np.float32))
                keep[sel] |= d < np.float32(f.get('udf_level_set', 2 * f['voxel_size']))
        return keep

    def extract_dual_mesh(self, mise_iter=0, grid_upsample=1, info=None, dual_graph='lattice'):
        has_mask = any(f.get('udf_feats') is not None for f in self.fields.values())
        if dual_graph == 'adaptive':          # the dual graph of the UNION hierarchy's flattened levels, blended field values
            from . import dual_adaptive
            return dual_adaptive.extract(self.voxel_size, [L.ijk for L in self.union.levels[:self.adaptive_depth]], lambda p: self.evaluate(p)[0],
                                         mise_iter, grid_upsample, mask_fn=(self.mask if has_mask else None), info=info)
        return meshing.extract(self.voxel_size, self.union.levels[0], lambda p: self.evaluate(p)[0], mise_iter, grid_upsample,
                               mask_fn=(self.mask if has_mask else None), info=info, coarser=self.union.levels[1:self.adaptive_depth])


OV_FLOOR = 1.0        # blend half-width floor, in coarsest voxels        
BAND_EXTRA = 1.5      # data margin beyond core +- ov, in coarsest voxels (= the support radius of the coarsest kernel); None = ov


def reconstruct_by_chunk(xyz, normal, sensor, chunk_size, overlap_ratio=0.05, preprocess_fn=None, voxel_size=0.1, depth=4,
                         min_points=8, ov_floor=None, band_extra=None, **kw):
    """``kw`` goes to oracle.pipeline.reconstruct (adaptive_depth, kernel_dim, hidden, tol, approx_kernel_grad,
    net_params, udf, ...).  ``preprocess_fn(xyz, normal, sensor) -> (xyz, normal, sensor)`` runs per chunk."""
    xyz = np.asarray(xyz, np.float32)
    lo = [float(v) for v in xyz.min(0)]
    hi = [float(v) for v in xyz.max(0)]
    grid = chunk_grid(lo, hi, chunk_size)
    wc = voxel_size * 2 ** (depth - 1)
    ov = max(overlap_ratio * chunk_size, (OV_FLOOR if ov_floor is None else ov_floor) * wc)
    be = BAND_EXTRA if band_extra is None else band_extra
    band = 2 * ov if be is None else ov + be * wc            # chunk c solves the points inside core_c +- band
    nchunk = grid[0] * grid[1] * grid[2]
    cores = {}
    for c in range(nchunk):
        cz, cy, cx = c % grid[2], (c // grid[2]) % grid[1], c // (grid[1] * grid[2])
        clo = [lo[0] + cx * chunk_size, lo[1] + cy * chunk_size, lo[2] + cz * chunk_size]
        cores[c] = (clo, [clo[a] + chunk_size for a in range(3)])
    counts = np.bincount(chunk_ids(xyz, lo, chunk_size, grid), minlength=nchunk)
    frame = Frame(voxel_size, depth, lo, grid, chunk_size, band)
    fields = {}
    for c in range(nchunk):
        if counts[c] == 0:
            continue
        clo, chi = cores[c]
        m = np.ones(xyz.shape[0], bool)
        for a in range(3):
            if grid[a] > 1:
                m &= (xyz[:, a] >= _f32(clo[a] - band)) & (xyz[:, a] < _f32(chi[a] + band))
        cx_, cn_, cs_ = xyz[m], (normal[m] if normal is not None else None), (sensor[m] if sensor is not None else None)
        if preprocess_fn is not None:
            cx_, cn_, cs_ = preprocess_fn(cx_, cn_, cs_)
        if cn_ is None:
            raise RuntimeError('oriented input required')
        if cx_.shape[0] < min_points:
            continue
        # preprocess_fn saw the chunk in the caller's coordinates; the solve happens in the chunk's slot of the exploded frame
        xs = (np.ascontiguousarray(cx_, np.float32) + frame.shift(c)[None]).astype(np.float32)
        fields[c] = pipeline.reconstruct(xs, np.ascontiguousarray(cn_, np.float32), voxel_size=voxel_size, depth=depth, **kw)
    return ChunkedField(fields, cores, ov, lo, chunk_size, grid, voxel_size, frame, adaptive_depth=kw.get('adaptive_depth', 1))
