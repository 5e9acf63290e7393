"""Spatial chunking (``chunk_size=``) and its multi-GPU sharding.

Reference behaviour (call sites; the implementation is in the absent wheel): ``reconstruct(xyz,
normal, detail_level=None, chunk_size=50.0)`` examples/recons_by_chunk.py:29; solved chunks are
parked on ``chunk_tmp_device`` (:27); "Tuning detail_level / voxel_size is not supported if
chunk_size is provided" NKSR-USAGE.md:137.  Spec:
  * the bounding box is cut into a grid of ``chunk_size`` cubes; chunk c solves the points inside
    core_c +- band (ov = max(overlap_ratio*chunk_size, OV_FLOOR coarsest voxels), band = ov + BAND_EXTRA coarsest voxels)
  * the global field is the partition-of-unity blend  f = sum_c w_c f_c / sum_c w_c  with
    w_c(x) = prod_axis ramp((x - (lo-ov)) / 2ov) * ramp(((hi+ov) - x) / 2ov)  (linear ramps of
    neighbouring chunks add up to 1 inside the 2*ov band; a chunk's weight vanishes ov inside
    its data boundary, so it never contributes where its hierarchy is truncated)
  * all chunks share ONE global voxel lattice (cells are floor(x / w) in global coordinates), so
    the union of the chunks' finest levels is a consistent dual grid; every dual cell is meshed by
    the rank that owns the chunk whose core contains the cell's base voxel centre.
Multi-GPU (one process per GPU): chunks are sharded over ranks (nksr_amd.dist) along a Morton curve; every rank is
given either the same full cloud or -- ``sharded_input=True`` -- only the points of its own chunks (+ band).  No
collective on the solve path, one all_gather of the chunk HALOS before meshing, one point-to-point gather of the
mesh pieces to rank 0 after it.
"""
import math

import torch

from . import dist as D
from .fields.base_field import BaseField, EvaluationResult
from .fields.kernel_field import KernelField
from .fields.mask_fields import LayerField, NeuralField
from .svh import SparseFeatureHierarchy


def chunk_grid(lo, hi, chunk_size):
    n = [max(1, int(math.ceil((hi[a] - lo[a]) / chunk_size - 1e-9))) for a in range(3)]
    return n


def _udf_levels(f):
    m = f.mask_field
    if not isinstance(m, NeuralField):
        return 0
    n = 0
    while n < f.svh.depth and n < len(m.features) and m.features[n] is not None:
        n += 1
    return n


def exchange_band(core, cidx3, grid, ov, w0):
    """Where OTHER ranks evaluate this chunk's field: along every split axis with a neighbouring chunk, from
    2.5 finest voxels inside the shared face (their one-ring of halo cells and its refined lattice) to ``ov``
    outside it (the end of the blend weight).  Returns [(axis, lo, hi), ...]; see pack_field(band=)."""
    clo, chi = core
    out = []
    for a in range(3):
        if grid[a] <= 1:
            continue
        if cidx3[a] > 0:
            out.append((a, clo[a] - ov, clo[a] + 2.5 * w0))
        if cidx3[a] < grid[a] - 1:
            out.append((a, chi[a] - 2.5 * w0, chi[a] + ov))
    return out


def pack_field(f, band=None):
    """KernelField (+ its UDF mask features, when the mask is a NeuralField) -> (int64 tensor, float32 tensor).
    ``band`` (exchange_band): keep only the voxels that can contribute to an evaluation inside the band -- at
    level d those whose centre lies within 2.5 w_d of it (B-spline support 1.5 w_d, trilinear feature
    stencil 1 w_d).  This is the "halo" payload of the rank exchange: evaluations
    inside the band are bit-identical to those of the full field."""
    svh = f.svh
    nu = _udf_levels(f)
    keep = [None] * svh.depth
    if band is not None:
        for d in range(svh.depth):
            g = svh.level(d)
            w = g.voxel_size
            m = torch.zeros(g.num_voxels, dtype=torch.bool, device=svh.device)
            for a, lo, hi in band:
                ca = (g.ijk[:, a].to(torch.float32) + 0.5) * w
                m |= (ca >= lo - 2.5 * w) & (ca <= hi + 2.5 * w)
            keep[d] = m
    sel = lambda t, d: t if keep[d] is None else t[keep[d]]
    off = svh.offsets
    ns = [int(svh.num_voxels(d) if keep[d] is None else keep[d].sum()) for d in range(svh.depth)]
    head = [svh.depth, f.kdim, int(f.approx_kernel_grad), nu] + ns
    ints = torch.cat([torch.tensor(head, dtype=torch.int64, device=svh.device)] + [sel(svh.level(d).keys, d) for d in range(svh.depth)])
    parts = [sel(f._feat[d], d).reshape(-1) for d in range(svh.depth)]
    parts += [sel(f.alpha[off[d]:off[d] + svh.num_voxels(d)], d) for d in range(svh.depth)]
    if nu:
        parts += [sel(f.mask_field.features[d], d).reshape(-1) for d in range(nu)]
        parts.append(torch.tensor([f.mask_field.level_set], dtype=torch.float32, device=svh.device))
    return ints, torch.cat(parts)


def unpack_field(ints, flts, voxel_size, interpolators, device):
    ints, flts = ints.to(device), flts.to(device)
    depth, kdim, approx, nu = int(ints[0]), int(ints[1]), bool(int(ints[2])), int(ints[3])
    ns = [int(v) for v in ints[4:4 + depth]]
    off = 4 + depth
    keys = []
    for n in ns:
        keys.append(ints[off:off + n].contiguous())
        off += n
    svh = SparseFeatureHierarchy(voxel_size, depth, device).build_from_keys(keys, sorted_unique=True)
    feats, fo = [], 0
    for n in ns:
        feats.append(flts[fo:fo + n * kdim].view(n, kdim).contiguous())
        fo += n * kdim
    fld = KernelField(svh, interpolators, feats, approx_kernel_grad=approx)
    fld.alpha = flts[fo:fo + sum(ns)].contiguous()
    fo += sum(ns)
    if nu:
        from .nn.network import UDFDecoder
        uf = [None] * depth
        for d in range(nu):
            uf[d] = flts[fo:fo + ns[d] * 8].view(ns[d], 8).contiguous()
            fo += ns[d] * 8
        mask = NeuralField(svh, UDFDecoder(), uf)
        mask.set_level_set(float(flts[fo]))
        fld.set_mask_field(mask)
    return fld


class ChunkUnionMask(BaseField):
    """Mask of a chunked field whose chunks carry NeuralField (UDF) masks: a vertex survives when any
    chunk that contributes to the blend there keeps it."""

    def __init__(self, multi):
        super().__init__(multi.svh)
        self.multi = multi

    def evaluate_mask(self, xyz_model):
        keep = torch.zeros(xyz_model.shape[0], dtype=torch.bool, device=xyz_model.device)
        for c in sorted(self.multi.fields):
            f = self.multi.fields[c]
            if f.mask_field is None:
                continue
            sel = torch.nonzero(self.multi._weight(c, xyz_model) > 0).reshape(-1)
            if sel.numel():
                keep[sel] |= f.mask_field.evaluate_mask(xyz_model[sel].contiguous())
        return keep

    def to_(self, device):
        return self


class MultiChunkField(BaseField):
    """Partition-of-unity blend of the chunk fields.  With one rank it holds every chunk and is valid everywhere.
    With several ranks a rank holds its own chunks in full and only the HALO of its spatial neighbours
    (chunking.exchange_band): ``evaluate_f`` is then exact inside the rank's own cores (+ the one-voxel ring it
    meshes) and must not be used elsewhere -- ``extract_dual_mesh`` respects that and gathers the pieces."""

    def __init__(self, fields, cores, ov, origin, chunk_size, grid, owner, rank, world_size, voxel_size, device):
        self.fields = fields              # {chunk id: KernelField}
        self.cores = cores                # {chunk id: (lo[3], hi[3])} model units
        self.ov = float(ov)
        self.origin, self.chunk_size, self.grid = origin, float(chunk_size), grid
        self.owner, self.rank, self.world_size = owner, rank, world_size
        keys = [f.svh.level(0).keys for f in fields.values() if f.svh.num_voxels(0) > 0]
        union = SparseFeatureHierarchy(voxel_size, 1, device)
        union.build_from_keys([torch.cat(keys) if keys else None])
        super().__init__(union)
        self.mask_field = LayerField(union, 1)
        if any(isinstance(f.mask_field, NeuralField) for f in fields.values()):
            self.mask_field = ChunkUnionMask(self)
        self.solve_info = {}

    # ---- blend weights ------------------------------------------------------------------------
    def _weight(self, c, xyz):
        lo, hi = self.cores[c]
        w = torch.ones(xyz.shape[0], dtype=torch.float32, device=xyz.device)
        for a in range(3):
            x = xyz[:, a]
            if self.grid[a] > 1:          # no ramp along an axis that is not split
                w = w * ((x - (lo[a] - self.ov)) / (2 * self.ov)).clamp(0, 1) * (((hi[a] + self.ov) - x) / (2 * self.ov)).clamp(0, 1)
        return w

    def _evaluate_f_model(self, xyz, grad, max_points=1 << 22):
        n = xyz.shape[0]
        num = torch.zeros(n, dtype=torch.float32, device=xyz.device)
        den = torch.zeros(n, dtype=torch.float32, device=xyz.device)
        gnum = torch.zeros((n, 3), dtype=torch.float32, device=xyz.device) if grad else None
        # a chunk's weight is supported on core +- ov: only queries whose home chunk lies within `reach`
        # chunks of it can see it.  Queries are binned by home chunk once (one sort), so every chunk
        # weighs its neighbourhood instead of the whole query set (O(27 n) instead of O(chunks * n)).
        nchunk = self.grid[0] * self.grid[1] * self.grid[2]
        binned = n > 0 and nchunk > 8
        if binned:
            home = self.chunk_of(xyz)
            order = torch.sort(home, stable=True).indices
            off = [0] + torch.cumsum(torch.bincount(home, minlength=nchunk), 0).tolist()
            reach = max(1, int(math.ceil(self.ov / self.chunk_size)))
        for c in sorted(self.fields):     # fixed order => identical arithmetic on every rank
            if binned:
                cz, cy, cx = c % self.grid[2], (c // self.grid[2]) % self.grid[1], c // (self.grid[1] * self.grid[2])
                segs = []
                for ax in range(max(cx - reach, 0), min(cx + reach, self.grid[0] - 1) + 1):
                    for ay in range(max(cy - reach, 0), min(cy + reach, self.grid[1] - 1) + 1):
                        z0, z1 = max(cz - reach, 0), min(cz + reach, self.grid[2] - 1)
                        h0 = (ax * self.grid[1] + ay) * self.grid[2] + z0      # z-neighbours are consecutive ids
                        if off[h0 + z1 - z0 + 1] > off[h0]:
                            segs.append(order[off[h0]:off[h0 + z1 - z0 + 1]])
                if not segs:
                    continue
                cand = torch.cat(segs) if len(segs) > 1 else segs[0]
                pts = xyz[cand]
            else:
                cand, pts = None, xyz
            w = self._weight(c, pts)
            sel = torch.nonzero(w > 0).reshape(-1)
            if sel.numel() == 0:
                continue
            res = self.fields[c]._evaluate_f_model(pts[sel].contiguous(), grad, max_points)
            tgt = sel if cand is None else cand[sel]
            num.index_add_(0, tgt, res.value * w[sel])
            den.index_add_(0, tgt, w[sel])
            if grad:   # the gradient of the weights is ignored (they are flat outside the seams)
                gnum.index_add_(0, tgt, res.gradient * w[sel, None])
        den = den.clamp_min(1e-20)
        return EvaluationResult(num / den, gnum / den[:, None] if grad else None)

    # ---- ownership of dual cells ----------------------------------------------------------------
    def chunk_of(self, xyz):
        idx = []
        for a in range(3):
            i = torch.floor((xyz[:, a] - self.origin[a]) / self.chunk_size).long().clamp(0, self.grid[a] - 1)
            idx.append(i)
        return (idx[0] * self.grid[1] + idx[1]) * self.grid[2] + idx[2]

    def base_cell_mask(self, ijk):
        if self.world_size == 1:
            return torch.ones(ijk.shape[0], dtype=torch.bool, device=ijk.device)
        centers = (ijk.to(torch.float32) + 0.5) * self.svh.voxel_size
        own = torch.tensor(self.owner, dtype=torch.long, device=ijk.device)
        return own[self.chunk_of(centers)] == self.rank

    def base_cell_halo_mask(self, ijk):
        """Owned cells plus one ring of neighbours: the MISE hanging-vertex rule needs to know whether
        the cells across a rank seam were refined, so they are evaluated (but not meshed) here too."""
        if self.world_size == 1:
            return torch.ones(ijk.shape[0], dtype=torch.bool, device=ijk.device)
        own = torch.tensor(self.owner, dtype=torch.long, device=ijk.device)
        w = self.svh.voxel_size
        centers = (ijk.to(torch.float32) + 0.5) * w
        # chunk index of centre - w / centre / centre + w along every split axis; only voxels next to a chunk
        # boundary (lo != hi on some axis) can see a different owner than their own chunk's
        split = [a for a in range(3) if self.grid[a] > 1]
        idx = {}
        for a in split:
            for k, off in ((0, -w), (1, 0.0), (2, w)):
                idx[(a, k)] = torch.floor((centers[:, a] + off - self.origin[a]) / self.chunk_size).long().clamp_(0, self.grid[a] - 1)

        def lin(sel, ks):
            out = torch.zeros(1, dtype=torch.long, device=ijk.device)
            for a in range(3):
                ia = idx[(a, ks[a])][sel] if a in split else 0
                out = out * self.grid[a] + ia
            return out

        every = slice(None)
        m = own[lin(every, (1, 1, 1))] == self.rank
        near = torch.zeros(ijk.shape[0], dtype=torch.bool, device=ijk.device)
        for a in split:
            near |= idx[(a, 0)] != idx[(a, 2)]
        sel = torch.nonzero(near).reshape(-1)
        if sel.numel():
            ms = m[sel]
            combos = [()]
            for a in range(3):
                combos = [c + (k,) for c in combos for k in ((0, 1, 2) if a in split else (1,))]
            for ks in combos:
                ms = ms | (own[lin(sel, ks)] == self.rank)
            m[sel] = ms
        return m

    def finalize_mesh(self, res):
        if self.world_size == 1:
            return res
        v, f = D.gather_meshes(res.v, res.f, res.edge_vkey, res.edge_axis)
        res.v, res.f = v, f
        res.c = self.texture_field.evaluate_color(v) if self.texture_field is not None else None
        return res

    def for_rank(self, rank, world_size, fields):
        """Same scene seen from another (simulated) rank holding ``fields`` -- test helper."""
        return MultiChunkField(fields, self.cores, self.ov, self.origin, self.chunk_size, self.grid, self.owner, rank,
                               world_size, self.svh.voxel_size, self.svh.device)

    def to_(self, device):
        for f in self.fields.values():
            f.to_(device)
        self.svh.to_(device)
        return self


OV_FLOOR = 1.0        # blend half-width floor, in coarsest voxels
BAND_EXTRA = 1.5      # data margin beyond core +- ov, in coarsest voxels (= the support radius of the coarsest kernel); None = ov
MIN_CHUNK_POINTS = 8


def chunk_geometry(hp, chunk_size, overlap_ratio):
    wc = hp.voxel_size * 2 ** (hp.tree_depth - 1)
    ov = max(overlap_ratio * chunk_size, OV_FLOOR * wc)
    band = 2 * ov if BAND_EXTRA is None else ov + BAND_EXTRA * wc
    return ov, band


def reconstruct_by_chunk(rec, xyz, normal, sensor, chunk_size, overlap_ratio, approx_kernel_grad, solver_max_iter,
                         solver_tol, fused_mode, preprocess_fn, sim=None, sharded_input=False, chunk_owner=None, chunk_bounds=None):
    """``sim=(rank, world_size)`` runs one simulated rank without a process group (tests).
    ``sharded_input``: every rank passes only ITS part of the cloud -- at least the points inside core +- band of the
    chunks it owns ("each rank receives only its chunks' points (+overlap)").  The chunk grid then
    comes from ``chunk_bounds`` = (lo[3], hi[3]) or from an all_reduce of the local bounding boxes, the per-core point
    counts from an all_reduce(MAX) (some rank holds every core completely), and ``chunk_owner`` (list, one rank per
    chunk) lets the caller that distributed the data dictate the ownership it assumed."""
    hp = rec.hparams
    dev = rec.device
    rank, ws = sim if sim is not None else D.world()
    collective = sharded_input and ws > 1 and sim is None
    from .density import bbox_center
    if xyz.shape[0] and (not bool(torch.isfinite(xyz).all())):
        raise RuntimeError('non-finite coordinates in the input')
    if chunk_bounds is not None:
        lo, hi = [float(v) for v in chunk_bounds[0]], [float(v) for v in chunk_bounds[1]]
    else:
        if xyz.shape[0]:
            lo_t, hi_t, _ = bbox_center(xyz)
        else:
            lo_t = torch.full((3,), float('inf'), device=dev)
            hi_t = -lo_t
        if collective:
            import torch.distributed as dist
            cd = D._comm_device(lo_t)
            lo_t, hi_t = lo_t.to(cd), hi_t.to(cd)
            dist.all_reduce(lo_t, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi_t, op=dist.ReduceOp.MAX)
        lo, hi = [float(v) for v in lo_t.tolist()], [float(v) for v in hi_t.tolist()]
    grid = chunk_grid(lo, hi, chunk_size)
    ov, band = chunk_geometry(hp, chunk_size, overlap_ratio)
    nchunk = grid[0] * grid[1] * grid[2]
    cores = {}
    for c in range(nchunk):
        cz, cy, cx = c % grid[2], (c // grid[2]) % grid[1], c // (grid[1] * grid[2])
        clo = [lo[0] + cx * chunk_size, lo[1] + cy * chunk_size, lo[2] + cz * chunk_size]
        cores[c] = (clo, [clo[a] + chunk_size for a in range(3)])
    # points per core in one pass (the load-balance weights; a chunk whose core is empty is skipped: the
    # bands around it are covered by its neighbours' weights)
    cid = torch.zeros(xyz.shape[0], dtype=torch.long, device=dev)
    for a in range(3):
        if grid[a] > 1:
            ia = torch.floor((xyz[:, a] - lo[a]) / chunk_size).long().clamp_(0, grid[a] - 1)
        else:
            ia = 0
        cid = cid * grid[a] + ia
    counts_t = torch.bincount(cid, minlength=nchunk)
    local_counts = counts_t.tolist()
    if collective:
        import torch.distributed as dist
        ct = counts_t.to(D._comm_device(counts_t))
        dist.all_reduce(ct, op=dist.ReduceOp.MAX)
        counts = ct.tolist()
    else:
        counts = local_counts
    if chunk_owner is not None:
        if len(chunk_owner) != nchunk:
            raise RuntimeError('chunk_owner has %d entries, the chunk grid %s has %d chunks' % (len(chunk_owner), grid, nchunk))
        owner = [int(o) for o in chunk_owner]
    else:
        owner = D.partition_chunks(nchunk, ws, counts, grid)
    def solve_chunk(c):
        """select -> (preprocess) -> solve one chunk on the CURRENT stream; returns (c, field or None)"""
        if sharded_input and local_counts[c] != counts[c]:
            raise RuntimeError('sharded input: rank %d owns chunk %d but holds %d of its %d core points' % (rank, c, local_counts[c], counts[c]))
        clo, chi = cores[c]
        m = None                                  # points inside core +- band (only along the split axes)
        for a in range(3):
            if grid[a] > 1:
                ma = (xyz[:, a] >= clo[a] - band) & (xyz[:, a] < chi[a] + band)
                m = ma if m is None else (m & ma)
        idx = torch.nonzero(m).reshape(-1) if m is not None else torch.arange(xyz.shape[0], device=dev)
        cx_, cn_, cs_ = xyz[idx].contiguous(), (normal[idx].contiguous() if normal is not None else None), \
            (sensor[idx].contiguous() if sensor is not None else None)
        if preprocess_fn is not None:
            try:
                cx_, cn_, cs_ = preprocess_fn(cx_, cn_, cs_)
            except RuntimeError as e:
                if 'need at least' in str(e):     # too few points for the normal estimator: treat the chunk as empty
                    return c, None
                raise
        if cn_ is None:
            raise RuntimeError('oriented input required (normal= or sensor= with a normal-estimating preprocess_fn)')
        if cx_.shape[0] < MIN_CHUNK_POINTS:       # a handful of stray points: nothing to solve, neighbours cover the band
            return c, None
        if not bool(torch.isfinite(cn_).all()):
            raise RuntimeError('non-finite normals in the input')
        fld = rec._reconstruct_single(cx_.contiguous(), cn_.to(torch.float32).contiguous(), approx_kernel_grad,
                                      solver_max_iter, solver_tol, fused_mode)
        fld.matrix = None                 # the CSR is not needed after the solve
        fld._fused_op = None
        if rec.chunk_tmp_device != dev and ws == 1 and sim is None:
            fld.to_(rec.chunk_tmp_device)  # reference semantics: park solved chunks elsewhere
        return c, fld

    jobs = [c for c in range(nchunk) if owner[c] == rank and counts[c] > 0]
    cs = getattr(rec, 'chunk_streams', None)
    nstreams = max(1, min(int(cs if cs is not None else (3 if fused_mode else 1)), len(jobs)))
    if nstreams > 1:
        # Chunks are independent: solve them on several HIP streams, one host thread each.  A chunk of a few 100 k points is a chain
        # of ~100 short kernels with host round trips for sizes in between (unique counts, nnz, PCG convergence checks); with
        # one stream the GPU idles through those, with several the gaps of one chunk are filled by another.  Results do not depend
        # on the interleaving (no float atomics, every chunk has its own buffers): bit-identical to the sequential run.
        from concurrent.futures import ThreadPoolExecutor
        torch.cuda.synchronize(dev)               # inputs (and anything still using memory the workers may be handed) are settled
        streams = [torch.cuda.Stream(device=dev) for _ in range(nstreams)]

        def worker(i):
            out = []
            torch.cuda.set_device(dev)
            with torch.cuda.stream(streams[i]):
                for c in jobs[i::nstreams]:
                    out.append(solve_chunk(c))
                streams[i].synchronize()
            return out
        with ThreadPoolExecutor(max_workers=nstreams) as ex:
            results = [r for part in ex.map(worker, range(nstreams)) for r in part]
    else:
        results = [solve_chunk(c) for c in jobs]
    local = {c: f for c, f in sorted(results, key=lambda r: r[0]) if f is not None}
    timing = {}
    for f in local.values():              # per-stage host time summed over chunks (they overlap when chunk_streams > 1)
        for k, v in getattr(f, 'timing', {}).items():
            timing[k] = timing.get(k, 0.0) + v
    rec.timing = timing
    if ws > 1 and sim is None:
        # the exchange carries the halo of every chunk (the voxels other ranks can touch), not the whole field; which
        # chunks were actually solved travels with it (a sparse chunk may have been skipped by its owner)
        def band_of(c):
            c3 = (c // (grid[1] * grid[2]), (c // grid[2]) % grid[1], c % grid[2])
            return exchange_band(cores[c], c3, grid, ov, hp.voxel_size)
        payload = D.exchange_payloads({c: pack_field(f, band_of(c)) for c, f in local.items()})
        solved = sorted(payload)
        # a rank only evaluates the blend inside its own cores (+ the halo ring it evaluates): it needs exactly the
        # chunks whose weight support (core +- ov) reaches there -- its spatial neighbours, not all N
        need = needed_chunks(cores, ov + 2.5 * hp.voxel_size, grid, [c for c in solved if owner[c] == rank], solved)
        fields = {c: (local[c] if c in local else unpack_field(payload[c][0], payload[c][1], hp.voxel_size,
                                                              rec.network.interpolators, dev)) for c in need}
    else:
        fields = local
        if rec.chunk_tmp_device != dev and sim is None:
            for f in fields.values():
                f.to_(dev)                # meshing runs on the GPU: bring the parked chunks back
    return MultiChunkField(fields, cores, ov, lo, chunk_size, grid, owner, rank, ws, hp.voxel_size, dev)


def needed_chunks(cores, margin, grid, owned, candidates):
    """Chunks whose core grown by ``margin`` touches the core of an owned chunk (owned included)."""
    out = set(owned)
    for c in candidates:
        if c in out:
            continue
        clo, chi = cores[c]
        for o in owned:
            olo, ohi = cores[o]
            if all(grid[a] == 1 or (clo[a] - margin < ohi[a] and chi[a] + margin > olo[a]) for a in range(3)):
                out.add(c)
                break
    return sorted(out)
