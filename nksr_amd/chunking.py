"""Spatial chunking (``chunk_size=``), batched over all chunks of a rank, and its multi-GPU sharding.

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
  * every dual cell is meshed by the rank that owns the chunk whose core contains the cell's base voxel centre.

The reference runs its chunks one after the other (examples/recons_by_chunk.py:26-29).  Here ALL chunks of a rank are
ONE launch sequence: chunk c is moved into its own aligned cube ("slot") of an EXPLODED FRAME -- x' = x + T_c, T_c a
whole number of coarsest voxels, slots far enough apart that no kernel support reaches from one into another -- and the
union of the translated clouds is reconstructed like a single cloud.  Its system is block diagonal by construction (one
block per chunk); the PCG keeps per-chunk scalars and stopping tests (nksr_segments_t), every summation order of the
operator is chunk-relative, so a chunk's solution does not depend on its batch mates, and the slot of a chunk depends on
its grid position only: the same chunk gives the same bits on 1, 2 or 8 ranks.  A slot is an aligned cube of the Morton
lattice, hence one contiguous key range per level: the chunks are the segments of the batch.  The blend evaluates the
batch field at x + T_c for every chunk c that weighs at x.

Multi-GPU (one process per GPU): chunks are sharded over ranks (nksr_amd.dist) along a Morton curve; every rank is
given either the same full cloud or -- ``sharded_input=True`` -- only the points of its own chunks (+ band).  No
collective on the solve path, one all_gather of the chunk HALOS before meshing, one point-to-point gather of the
mesh pieces to rank 0 after it.
"""
import contextlib
import os
import ctypes as C
import math
import weakref

import numpy as np
import torch

from . import dist as D
from . import ops
from ._lib import ChunkGridT, call, ptr, stream
from .fields.base_field import BaseField, EvaluationResult
from .fields.kernel_field import KernelField, Segments
from .fields.mask_fields import LayerField, NeuralField
from .normals import TooFewPoints as ChunkTooSmall      # a normal-estimating preprocess_fn on a chunk with fewer points than k: chunk skipped
from .svh import SparseFeatureHierarchy


@contextlib.contextmanager
def borrowed(obj, device):
    """A PARKED field (its tensors live on ``chunk_tmp_device`` / were moved by ``to_('cpu')``: NKSR-USAGE.md:150-167) made usable on
    ``device`` for the duration of the block: ``to_()`` replaces the object's tensors by copies on ``device``; on exit the references
    to the parked tensors are put back and the copies die.  Nothing travels back -- evaluation and meshing do not change a field.
    ``obj``: a KernelField (with its hierarchy and mask) or a SparseFeatureHierarchy."""
    device = torch.device(device)

    def _norm(d):      # 'cuda' names the CURRENT device: compare full (type, index) pairs -- a field parked on another GPU is NOT resident
        d = torch.device(d)
        return (d.type, d.index if d.index is not None else (torch.cuda.current_device() if d.type == 'cuda' and torch.cuda.is_available() else 0))
    if _norm(obj.device) == _norm(device):
        yield obj
        return
    objs = [obj]
    for o in (getattr(obj, 'svh', None), getattr(obj, 'mask_field', None), getattr(getattr(obj, 'mask_field', None), 'svh', None)):
        if o is not None and all(o is not q for q in objs):
            objs.append(o)
    saved = [(o, dict(o.__dict__)) for o in objs]
    try:
        obj.to_(device)
        yield obj
    finally:
        for o, d in saved:
            o.__dict__.clear()
            o.__dict__.update(d)


def spill_to_disk(field, directory):
    """A field PARKED on the host (``to_('cpu')``) moved on to DISK: every tensor of the field, its hierarchy and its mask is replaced
    by a file-backed copy (``torch.from_file(..., shared=True)``, one file per tensor under ``directory``, unlinked at once: the
    mapping keeps the blocks until the tensor dies, nothing is left behind).  The host then holds reclaimable page cache instead of
    anonymous memory, and ``borrowed()`` pages a part in when evaluation / meshing visits it -- the out-of-core flow of
    NKSR-USAGE.md:150-167 for scenes whose solved chunks exceed host memory too.  Returns the bytes written."""
    import os
    import uuid
    os.makedirs(directory, exist_ok=True)
    total = 0
    seen = set()

    def move(t):
        nonlocal total
        if not torch.is_tensor(t) or t.device.type != 'cpu' or t.numel() == 0 or t.dtype == torch.bool:
            return t
        path = os.path.join(directory, 'nksr_spill_%s.bin' % uuid.uuid4().hex)
        flat = torch.from_file(path, shared=True, size=t.numel(), dtype=t.dtype)
        flat.copy_(t.reshape(-1))
        os.unlink(path)
        total += t.numel() * t.element_size()
        return flat.view(t.shape)

    def walk(o):
        if o is None or id(o) in seen or not hasattr(o, '__dict__'):
            return
        seen.add(id(o))
        for k, v in list(o.__dict__.items()):
            if torch.is_tensor(v):
                o.__dict__[k] = move(v)
            elif isinstance(v, (list, tuple)) and v and all(torch.is_tensor(x) or x is None for x in v):
                o.__dict__[k] = type(v)(move(x) for x in v)
            elif isinstance(v, (list, tuple)):
                for x in v:
                    if type(x).__module__.startswith('nksr_amd'):
                        walk(x)
            elif type(v).__module__.startswith('nksr_amd') and not isinstance(v, type):
                walk(v)
    if field.device.type != 'cpu':
        raise RuntimeError('spill_to_disk: park the field on the host first (field.to_("cpu"))')
    walk(field)
    return total


def chunk_grid(lo, hi, chunk_size):
    n = [max(1, int(math.ceil((hi[a] - lo[a]) / chunk_size - 1e-9))) for a in range(3)]
    return n


OV_FLOOR = 1.0        # blend half-width floor, in coarsest voxels
BAND_EXTRA = 1.5      # data margin beyond core +- ov, in coarsest voxels (= the support radius of the coarsest kernel); None = ov
MIN_CHUNK_POINTS = 8
SLOT_GAP = 6          # empty coarsest voxels between the data of two slots (kernel support 1.5 + structure dilation 1, both sides, + slack)


def chunk_geometry(hp, chunk_size, overlap_ratio):
    wc = hp.voxel_size * 2 ** (hp.tree_depth - 1)
    ov = max(overlap_ratio * chunk_size, OV_FLOOR * wc)
    band = 2 * ov if BAND_EXTRA is None else ov + BAND_EXTRA * wc
    return ov, band


class ChunkFrame:
    """Geometry of the exploded frame.  Chunk (cx, cy, cz) of the grid owns the cube [s * 2^S, (s + 1) * 2^S)^3 of the finest
    lattice, s = (cx, cy, cz) - grid // 2 (centred: small coordinates keep fp32 resolution); its data is translated by
    T_c = slot origin + low - a_c, a_c = the chunk's data corner rounded down to a multiple of 2^depth finest voxels, low = 4
    coarsest voxels -- a whole number of voxels at EVERY level, so the chunk's own lattice is kept.
    2^S >= low + data extent + alignment slack + SLOT_GAP / 2 coarsest voxels."""

    def __init__(self, voxel_size, depth, lo, grid, chunk_size, band):
        self.w0, self.depth = float(voxel_size), int(depth)
        self.lo, self.grid, self.chunk_size, self.band = [float(v) for v in lo], [int(g) for g in grid], float(chunk_size), float(band)
        self.align = 1 << self.depth
        # a chunk's voxels reach up to one coarsest voxel beyond its points (structure dilation) and its kernels 1.5 more: the data
        # sits `low` finest voxels (4 coarsest, a multiple of the alignment) inside its slot and ends >= SLOT_GAP / 2 coarsest
        # voxels before the slot's end, so that ALL voxels of a chunk lie inside its slot (= its Morton key range)
        self.low = 2 * self.align
        ext = int(math.ceil((self.chunk_size + 2 * self.band) / self.w0)) + 2
        need = self.low + ext + self.align + ((SLOT_GAP // 2) << (self.depth - 1))
        S = self.depth
        while (1 << S) < need:
            S += 1
        self.S = S
        self.half = [g // 2 for g in self.grid]
        reach = max(max(self.grid[a] - self.half[a], self.half[a]) for a in range(3)) << S
        if reach >= (1 << 20) - (1 << S):
            raise RuntimeError('chunk grid %s with %d-voxel slots does not fit the 2^20 lattice: use a larger chunk_size' % (self.grid, 1 << S))
        # data must stay inside [usable_lo, usable_hi) of its slot (model units)
        self.usable_lo = (self.low - 1) * self.w0
        self.usable_hi = ((1 << S) - ((SLOT_GAP // 2) << (self.depth - 1))) * self.w0

    def chunk3(self, c):
        g = self.grid
        return (c // (g[1] * g[2]), (c // g[2]) % g[1], c % g[2])

    def slot_origin(self, c):
        c3 = self.chunk3(c)
        return [(c3[a] - self.half[a]) << self.S for a in range(3)]

    def shift_cells(self, c):
        c3 = self.chunk3(c)
        out = []
        for a in range(3):
            data_lo = self.lo[a] + c3[a] * self.chunk_size - self.band if self.grid[a] > 1 else self.lo[a]
            corner = int(math.floor(math.floor(data_lo / self.w0) / self.align)) * self.align
            out.append(((c3[a] - self.half[a]) << self.S) + self.low - corner)
        return out

    def shift(self, c):
        """T_c in model units, fp32 (the product of the integer voxel count and the voxel size, rounded once)."""
        return np.asarray([np.float32(t * self.w0) for t in self.shift_cells(c)], np.float32)

    def key_range(self, c):
        o = self.slot_origin(c)
        k = D._morton3(o[0] + (1 << 20), o[1] + (1 << 20), o[2] + (1 << 20))
        return k, k + (1 << (3 * self.S))


# ---- per-chunk payloads (rank exchange, save_field) ---------------------------------------------------------------------
def _udf_levels(f):
    m = f.mask_field
    if not isinstance(m, NeuralField):
        return 0
    n = 0
    while n < f.svh.depth and n < len(m.features) and m.features[n] is not None:
        n += 1
    return n


def halo_inner(voxel_size, adaptive_depth=1, dual_graph='lattice'):
    """How far INSIDE a neighbour's core a rank evaluates the blend (model units).  Lattice mesher: 2.5 finest voxels (the one-ring
    of halo cells and its refined lattice).  Adaptive dual graph: the hexahedra around a rank's own octree corners reach one leaf
    across the seam and the MISE split of that leaf looks one more leaf out -- leaves up to 2^(A-1) voxels wide, the rings of
    the finer MISE rounds half as deep each: 3 * 2^(A-1) + 1 voxels cover any mise_iter."""
    if dual_graph != 'adaptive':
        return 2.5 * voxel_size
    return (3.0 * (1 << (max(1, int(adaptive_depth)) - 1)) + 1.0) * voxel_size


def exchange_band(core, cidx3, grid, ov, w0, inner=None):
    """Where OTHER ranks evaluate this chunk's field: along every split axis with a neighbouring chunk, from
    ``inner`` (default 2.5 finest voxels: their one-ring of halo cells and its refined lattice; halo_inner) inside the shared
    face to ``ov`` outside it (the end of the blend weight).  Returns [(axis, lo, hi), ...]; see pack_field(band=)."""
    clo, chi = core
    inner = 2.5 * w0 if inner is None else float(inner)
    out = []
    for a in range(3):
        if grid[a] <= 1:
            continue
        if cidx3[a] > 0:
            out.append((a, clo[a] - ov, clo[a] + inner))
        if cidx3[a] < grid[a] - 1:
            out.append((a, chi[a] - inner, chi[a] + ov))
    return out


def _pack(svh, kdim, approx, feat, alpha, udf_feats, level_set, ranges, band, shift):
    """Payload of the voxels ``ranges[d] = (lo, hi)`` of a hierarchy (one chunk of a batch, or a whole field)."""
    depth = svh.depth
    nu = len(udf_feats)
    off = svh.offsets
    sels, ns = [], []
    for d in range(depth):
        g = svh.level(d)
        lo, hi = ranges[d]
        if band is None:
            sels.append(slice(lo, hi))
            ns.append(hi - lo)
            continue
        w = g.voxel_size
        m = torch.zeros(hi - lo, dtype=torch.bool, device=svh.device)
        for a, blo, bhi in band:
            ca = (g.ijk[lo:hi, a].to(torch.float32) + 0.5) * w - float(shift[a])
            m |= (ca >= blo - 2.5 * w) & (ca <= bhi + 2.5 * w)
        idx = torch.nonzero(m).reshape(-1) + lo
        sels.append(idx)
        ns.append(int(idx.numel()))
    head = [depth, kdim, int(approx), nu] + ns
    ints = torch.cat([torch.tensor(head, dtype=torch.int64, device=svh.device)] + [svh.level(d).keys[sels[d]] for d in range(depth)])
    parts = [feat[d][sels[d]].reshape(-1) for d in range(depth)]
    for d in range(depth):
        s = sels[d]
        parts.append(alpha[off[d] + s.start:off[d] + s.stop] if isinstance(s, slice) else alpha[s + off[d]])
    if nu:
        parts += [udf_feats[d][sels[d]].reshape(-1) for d in range(nu)]
        parts.append(torch.tensor([level_set], dtype=torch.float32, device=svh.device))
    return ints, torch.cat(parts)


def pack_field(f, band=None, shift=None):
    """KernelField (+ its UDF mask features, when the mask is a NeuralField) -> (int64 tensor, float32 tensor).
    ``band`` (exchange_band, GLOBAL coordinates; ``shift`` = the translation of the field's frame, default the field's
    ``chunk_shift`` or 0): keep only the voxels that can contribute to an evaluation inside the band -- at level d those whose
    centre lies within 2.5 w_d of it (B-spline support 1.5 w_d, trilinear feature stencil 1 w_d).  This is the "halo" payload of
    the rank exchange: evaluations inside the band are bit-identical to those of the full field."""
    svh = f.svh
    nu = _udf_levels(f)
    if shift is None:
        shift = getattr(f, 'chunk_shift', (0.0, 0.0, 0.0))
    return _pack(svh, f.kdim, f.approx_kernel_grad, f._feat, f.alpha, [f.mask_field.features[d] for d in range(nu)],
                 f.mask_field.level_set if nu else 0.0, [(0, svh.num_voxels(d)) for d in range(svh.depth)], band, shift)


def _payload_heads(ints_list):
    """The headers (4 + at most 6 level counts) of several payloads in ONE host read."""
    if not ints_list:
        return []
    return torch.stack([torch.nn.functional.pad(i[:10], (0, 10 - min(10, int(i.numel())))) for i in ints_list]).tolist()


def _parse_payload(ints, flts, head=None):
    if head is None:
        head = _payload_heads([ints])[0]            # (was one host read per header entry)
    depth, kdim, approx, nu = int(head[0]), int(head[1]), bool(int(head[2])), int(head[3])
    ns = [int(v) for v in head[4:4 + depth]]
    off = 4 + depth
    keys = []
    for n in ns:
        keys.append(ints[off:off + n])
        off += n
    feats, fo = [], 0
    for n in ns:
        feats.append(flts[fo:fo + n * kdim].view(n, kdim))
        fo += n * kdim
    alphas = []
    for n in ns:
        alphas.append(flts[fo:fo + n])
        fo += n
    uf, level_set = [], 0.0
    if nu:
        for d in range(nu):
            uf.append(flts[fo:fo + ns[d] * 8].view(ns[d], 8))
            fo += ns[d] * 8
        level_set = float(flts[fo])
    return dict(depth=depth, kdim=kdim, approx=approx, nu=nu, ns=ns, keys=keys, feats=feats, alphas=alphas, udf=uf, level_set=level_set)


def fields_from_payloads(payloads, voxel_size, interpolators, device):
    """ONE KernelField from the payloads of several chunks -- [(key_lo of the chunk's slot, ints, flts), ...]; the slots are
    disjoint key ranges, so concatenating the chunks in slot order gives every level in canonical (ascending key) order."""
    payloads = [(k, i.to(device), f.to(device)) for k, i, f in sorted(payloads, key=lambda p: p[0])]
    heads = _payload_heads([i for _, i, _ in payloads])
    ps = [_parse_payload(i, f, h) for (_, i, f), h in zip(payloads, heads)]
    depth, kdim, approx, nu = ps[0]['depth'], ps[0]['kdim'], ps[0]['approx'], max(p['nu'] for p in ps)
    keys = [torch.cat([p['keys'][d] for p in ps]).contiguous() for d in range(depth)]
    svh = SparseFeatureHierarchy(voxel_size, depth, device).build_from_keys(keys, sorted_unique=True)
    feats = [torch.cat([p['feats'][d] for p in ps]).contiguous() for d in range(depth)]
    fld = KernelField(svh, interpolators, feats, approx_kernel_grad=approx)
    fld.alpha = torch.cat([p['alphas'][d] for d in range(depth) for p in ps]).contiguous()
    if nu:
        from .nn.network import UDFDecoder
        uf = [None] * depth
        for d in range(nu):
            uf[d] = torch.cat([p['udf'][d] if d < p['nu'] else torch.zeros((p['ns'][d], 8), device=device) for p in ps]).contiguous()
        mask = NeuralField(svh, UDFDecoder(), uf)
        mask.set_level_set(next(p['level_set'] for p in ps if p['nu']))
        fld.set_mask_field(mask)
    return fld


def unpack_field(ints, flts, voxel_size, interpolators, device):
    return fields_from_payloads([(0, ints, flts)], voxel_size, interpolators, device)


class ChunkPart:
    """A KernelField in the exploded frame holding the chunks ``ids`` (ascending slot key), whole or as halos."""

    def __init__(self, field, ids, frame, solved=True):
        self.field, self.ids, self.frame, self.solved = field, list(ids), frame, solved
        self._ranges = None

    def ranges(self):
        """{chunk: [(lo, hi) per level]} voxel index ranges (one host read)."""
        if self._ranges is None:
            svh = self.field.svh
            kr = [self.frame.key_range(c) for c in self.ids]
            klo = torch.tensor([k[0] for k in kr], dtype=torch.int64, device=svh.device)
            khi = torch.tensor([k[1] for k in kr], dtype=torch.int64, device=svh.device)
            lo = torch.stack([torch.searchsorted(svh.level(d).keys, klo >> (3 * d)) for d in range(svh.depth)], 1).tolist()
            hi = torch.stack([torch.searchsorted(svh.level(d).keys, khi >> (3 * d)) for d in range(svh.depth)], 1).tolist()
            self._ranges = {c: [(lo[i][d], hi[i][d]) for d in range(svh.depth)] for i, c in enumerate(self.ids)}
        return self._ranges

    def pack_chunk(self, c, band=None):
        f = self.field
        nu = _udf_levels(f)
        return _pack(f.svh, f.kdim, f.approx_kernel_grad, f._feat, f.alpha, [f.mask_field.features[d] for d in range(nu)],
                     f.mask_field.level_set if nu else 0.0, self.ranges()[c], band, self.frame.shift(c))

    def pack_halos(self, bands):
        """``{c: (ints, flts)}`` for every chunk of the part -- exactly what ``pack_chunk(c, bands[c])`` returns, made for all chunks at
        once: one mask and one compaction per LEVEL instead of one per chunk and level (8 chunks x 5 levels of small launches and
        host syncs were 11 ms of a rank's 80 ms at 8 ranks, tools/prof_rank_tail.py)."""
        f = self.field
        svh, dev, depth = f.svh, f.svh.device, f.svh.depth
        nu = _udf_levels(f)
        ids, nc = self.ids, len(self.ids)
        if nc == 0:
            return {}
        kr = [self.frame.key_range(c) for c in ids]
        klo = torch.tensor([k[0] for k in kr], dtype=torch.int64, device=dev)
        shift = torch.from_numpy(np.stack([np.asarray(self.frame.shift(c), np.float32) for c in ids])).to(dev)      # [nc, 3]
        # band table: up to two intervals per axis (the faces shared with the chunk before / after); none = an empty interval
        blo = np.full((nc, 3, 2), np.inf)
        bhi = np.full((nc, 3, 2), -np.inf)
        for i, c in enumerate(ids):
            used = [0, 0, 0]
            for a, lo_, hi_ in bands[c]:
                blo[i, a, used[a]], bhi[i, a, used[a]] = lo_, hi_
                used[a] += 1
        off = svh.offsets
        sel, cnt = [], []
        for d in range(depth):
            g = svh.level(d)
            if g.num_voxels == 0:
                sel.append(torch.zeros(0, dtype=torch.long, device=dev))
                cnt.append(torch.zeros(nc, dtype=torch.long, device=dev))
                continue
            w = g.voxel_size
            # (thresholds in double, compared in fp32, as pack_field does with Python scalars; csrc/chunks.hip k_halo_band_flags)
            tlo = torch.from_numpy((blo - 2.5 * w).astype(np.float32)).to(dev)
            thi = torch.from_numpy((bhi + 2.5 * w).astype(np.float32)).to(dev)
            seg = torch.empty(g.num_voxels, dtype=torch.int32, device=dev)
            flags = torch.empty(g.num_voxels, dtype=torch.int32, device=dev)
            call('nksr_halo_band_flags', ptr(g.keys), ptr(g.ijk), g.num_voxels, ptr((klo >> (3 * d)).contiguous()), nc, ptr(shift), ptr(tlo), ptr(thi),
                 float(w), ptr(seg), ptr(flags), stream())
            idx = ops.compact(flags).long()
            sel.append(idx)
            cnt.append(torch.bincount(seg[idx].long(), minlength=nc))
        counts = torch.stack(cnt, 1).tolist()                                   # [nc][depth]   (one host read)
        keys = [svh.level(d).keys[sel[d]] for d in range(depth)]
        feats = [f._feat[d][sel[d]] for d in range(depth)]
        alphas = [f.alpha[sel[d] + off[d]] for d in range(depth)]
        udf = [f.mask_field.features[d][sel[d]] for d in range(nu)]
        heads = torch.tensor([[depth, f.kdim, int(f.approx_kernel_grad), nu] + counts[i] for i in range(nc)], dtype=torch.int64, device=dev)
        tail = [torch.tensor([f.mask_field.level_set], dtype=torch.float32, device=dev)] if nu else []
        out, o = {}, [0] * depth
        for i, c in enumerate(ids):
            sl = [slice(o[d], o[d] + counts[i][d]) for d in range(depth)]
            ints = torch.cat([heads[i]] + [keys[d][sl[d]] for d in range(depth)])
            parts = [feats[d][sl[d]].reshape(-1) for d in range(depth)] + [alphas[d][sl[d]] for d in range(depth)]
            parts += [udf[d][sl[d]].reshape(-1) for d in range(nu)] + tail
            out[c] = (ints, torch.cat(parts))
            o = [o[d] + counts[i][d] for d in range(depth)]
        return out

    def chunk_view(self, c, interpolators):
        """Chunk c as a KernelField of its own (exploded frame): tests, save_field, simulated ranks."""
        ints, flts = self.pack_chunk(c)
        g = unpack_field(ints, flts, self.field.svh.voxel_size, interpolators, self.field.device)
        g.chunk_shift = tuple(float(v) for v in self.frame.shift(c))
        if g.mask_field is None:
            g.set_mask_field(LayerField(g.svh, getattr(self.field.mask_field, 'adaptive_depth', 1)))
        g.meshing_depth = getattr(self.field, 'meshing_depth', 1)
        info = self.field.solve_info
        g.solve_info = {}
        if self.solved and info:
            si = info.get('segment_info')
            i = self.ids.index(c)
            g.solve_info = {'M': int(g.svh.num_unknowns), 'nnz': 0, 'fused': True,
                            'iters': int(si[i, 0]) if si is not None else info.get('iters'),
                            'rel_residual': float(si[i, 1]) if si is not None else info.get('rel_residual')}
        return g


def chunk_grid_struct(origin, grid, chunk_size, sel_band, w_band, device, shift=None):
    """nksr_chunk_grid_t (include/nksr_hip.h) + the device arrays it points to (keep both alive).  Bounds are rounded to fp32 once,
    on the host: lo_sel/hi_sel = core -+ sel_band (membership of the solve), lo_w/hi_w = core -+ w_band (blend ramps)."""
    G = ChunkGridT()
    keep = []
    # candidate window of a point: its home chunk +- reach.  floor + 1, not ceil: the kernel finds the home chunk with x * (1 / chunk_size),
    # the host bounds with origin + j * chunk_size -- at an exact chunk boundary the two may differ by one
    reach = 1
    for b in (sel_band, w_band):
        if b is not None:
            reach = max(reach, int(math.floor(b / chunk_size)) + 1)
    if reach > 4:
        raise RuntimeError('chunk_size %g is too small for this hierarchy: a chunk is solved on its core + a band of %g (overlap + 1.5 coarsest '
                           'voxels), which must stay below 4 chunk sizes -- use chunk_size >= %g' % (chunk_size, max(b for b in (sel_band, w_band) if b is not None),
                                                                                              max(b for b in (sel_band, w_band) if b is not None) / 3.9))
    for a in range(3):
        G.grid[a] = int(grid[a])
        G.origin[a] = float(origin[a])
        if grid[a] <= 1:
            continue
        for band, lo_name, hi_name in ((sel_band, 'lo_sel', 'hi_sel'), (w_band, 'lo_w', 'hi_w')):
            if band is None:
                continue
            lo_t = torch.tensor([np.float32(origin[a] + j * chunk_size - band) for j in range(grid[a])], dtype=torch.float32, device=device)
            hi_t = torch.tensor([np.float32(origin[a] + j * chunk_size + chunk_size + band) for j in range(grid[a])], dtype=torch.float32, device=device)
            keep += [lo_t, hi_t]
            getattr(G, lo_name)[a] = ptr(lo_t)
            getattr(G, hi_name)[a] = ptr(hi_t)
    G.reach = reach
    G.inv_cs = float(np.float32(1.0) / np.float32(chunk_size))
    if w_band is not None:
        G.inv_2ov = float(np.float32(1.0) / np.float32(2 * w_band))
    if shift is not None:
        keep.append(shift)
        G.shift = ptr(shift)
    return G, keep


class _ChunkViews:
    """``multi.fields``: per-chunk KernelFields, built on demand from the batch (the batch is what evaluates).  Holds the parts, not
    the MultiChunkField: a reference back to it would be a cycle, and the field (GBs of device memory) would then live until
    the cyclic collector happens to run instead of until its last reference goes."""

    def __init__(self, parts, part_of, interpolators):
        self._parts, self._part_of, self._interp, self._cache = parts, part_of, interpolators, {}

    def _ids(self):
        return sorted(self._part_of)

    def __iter__(self):
        return iter(self._ids())

    def __len__(self):
        return len(self._part_of)

    def __contains__(self, c):
        return c in self._part_of

    def keys(self):
        return self._ids()

    def __getitem__(self, c):
        if c not in self._cache:
            self._cache[c] = self._parts[self._part_of[c]].chunk_view(c, self._interp)
        return self._cache[c]

    def values(self):
        return [self[c] for c in self._ids()]

    def items(self):
        return [(c, self[c]) for c in self._ids()]


class ChunkUnionMask(BaseField):
    """Mask of a chunked field whose chunks carry NeuralField (UDF) masks: a vertex survives when any
    chunk that contributes to the blend there keeps it."""

    def __init__(self, multi):
        super().__init__(multi.svh)
        self._multi = weakref.ref(multi)          # the field owns its mask, not the other way round (no reference cycle)

    def evaluate_mask(self, xyz_model):
        m = self._multi()
        if m is None:
            raise RuntimeError('the chunked field this mask belongs to has been released')
        xyz_model = xyz_model.contiguous()
        keep = torch.zeros(xyz_model.shape[0], dtype=torch.bool, device=xyz_model.device)
        if xyz_model.shape[0] == 0:
            return keep
        _, q, cid, _, xq = m._pairs(xyz_model)
        pl = m._part_lut[cid.long()] if len(m.parts) > 1 else None
        for pi, part in enumerate(m.parts):
            if part.field.mask_field is None or not isinstance(part.field.mask_field, NeuralField):
                continue
            s = torch.nonzero(pl == pi).reshape(-1) if pl is not None else None
            qq, xx = (q, xq) if s is None else (q[s], xq[s].contiguous())
            if qq.numel():
                with borrowed(part.field, m.home) as pf:
                    kq = pf.mask_field.evaluate_mask(xx)
                keep[qq[kq]] = True          # a query may appear once per chunk: "any chunk keeps it"
        return keep

    def to_(self, device):
        return self


class MultiChunkField(BaseField):
    """Partition-of-unity blend of the chunk fields.  With one rank it holds every chunk and is valid everywhere.
    With several ranks a rank holds its own chunks in full and only the HALO of its spatial neighbours
    (chunking.exchange_band): ``evaluate_f`` is then exact inside the rank's own cores (+ the one-voxel ring it
    meshes) and must not be used elsewhere -- ``extract_dual_mesh`` respects that and gathers the pieces."""

    def __init__(self, parts, cores, ov, origin, chunk_size, grid, owner, rank, world_size, frame, interpolators, device, distributed=False,
                 adaptive_depth=1, halo_inner=None):
        self.parts = [p for p in parts if p.ids]
        self.halo_inner = halo_inner          # (model units; None: the lattice mesher's 2.5 voxels -- chunking.halo_inner)
        self.cores = cores                # {chunk id: (lo[3], hi[3])} model units
        self.ov = float(ov)
        self.origin, self.chunk_size, self.grid = origin, float(chunk_size), grid
        self.owner, self.rank, self.world_size = owner, rank, world_size
        self.frame, self.interpolators, self.distributed = frame, interpolators, bool(distributed)
        self.part_of = {c: i for i, p in enumerate(self.parts) for c in p.ids}
        nchunk = grid[0] * grid[1] * grid[2]
        lut = torch.full((nchunk,), -1, dtype=torch.long)
        shifts = np.zeros((nchunk, 3), np.float32)
        cells = np.zeros((nchunk, 3), np.int32)
        for c, i in self.part_of.items():
            lut[c] = i
            shifts[c] = frame.shift(c)
            cells[c] = frame.shift_cells(c)
        self._part_lut = lut.to(device)
        self._shift = torch.from_numpy(shifts).to(device)
        self._chunk_flag = self._part_lut.to(torch.int32)
        self._cgrid, self._cgrid_keep = chunk_grid_struct(origin, grid, chunk_size, None, self.ov, device, shift=self._shift)
        # union of the chunks' voxels on the GLOBAL lattice (integer translation back; T_c is a whole number of voxels at every
        # level): the dual grid that is meshed -- the finest level and, below adaptive_depth, the coarser ones (LayerField(dec_svh,
        # adaptive_depth), models/nksr_net.py:132; 2 in the carla preset, configs/carla/train.yaml:6)
        nlev = max(1, min(int(adaptive_depth), frame.depth))
        keys = [[] for _ in range(nlev)]
        for p in self.parts:
            kr = torch.tensor([frame.key_range(c)[0] for c in p.ids], dtype=torch.int64, device=device)
            sc_all = torch.from_numpy(cells[p.ids]).to(device)
            for d in range(min(nlev, p.field.svh.depth)):
                g = p.field.svh.level(d)
                if g.num_voxels == 0:
                    continue
                gk, gi = g.keys.to(device), g.ijk.to(device)              # (a parked part: its finest keys visit the GPU for this)
                seg = torch.bucketize(gk, kr >> (3 * d), right=True) - 1
                ijk = (gi - (sc_all[seg] >> d)).contiguous()
                k = torch.empty(ijk.shape[0], dtype=torch.int64, device=device)
                call('nksr_encode_keys', ptr(ijk), ijk.shape[0], d, ptr(k), stream())
                keys[d].append(k)
        union = SparseFeatureHierarchy(frame.w0, nlev, device)
        union.build_from_keys([torch.cat(k) if k else None for k in keys])
        super().__init__(union)
        self.meshing_depth = nlev
        self.mask_field = LayerField(union, nlev)
        if any(isinstance(p.field.mask_field, NeuralField) for p in self.parts):
            self.mask_field = ChunkUnionMask(self)
        self.solve_info = {}
        self.fields = _ChunkViews(self.parts, self.part_of, interpolators)
        self.home = torch.device(device)           # where evaluation and meshing run, wherever the parts are parked

    def chunk_infos(self):
        """[{'chunk', 'M', 'iters', 'rel_residual'}] of the chunks solved here (one host read per part)."""
        out = []
        for p in self.parts:
            info = p.field.solve_info
            if not p.solved or not info:
                continue
            si = info.get('segment_info')
            si = si.tolist() if si is not None else [[info['iters'], info['rel_residual']]]
            rg = p.ranges()
            for i, c in enumerate(p.ids):
                out.append({'chunk': c, 'M': sum(h - l for l, h in rg[c]), 'iters': int(si[i][0]), 'rel_residual': float(si[i][1])})
        return out

    # ---- blend ------------------------------------------------------------------------------------
    def _pairs(self, xyz):
        """(offsets [n + 1] int32, query index, chunk, weight, translated position) of every (query, chunk) with a positive blend
        weight, the pairs of a query in ASCENDING chunk order -- the fixed summation order of the blend (csrc/chunks.hip).  A
        chunk's weight is supported on core +- ov, so only the chunks around a query's home chunk are candidates."""
        n = xyz.shape[0]
        dev = xyz.device
        counts = torch.empty(n + 1, dtype=torch.int32, device=dev)
        counts[n:] = 0
        call('nksr_chunk_pair_counts', C.byref(self._cgrid), 1, ptr(xyz), n, ptr(self._chunk_flag), ptr(counts), stream())
        offs = ops.exclusive_sum_i32(counts)
        m = int(offs[n].item())
        q = torch.empty(m, dtype=torch.int64, device=dev)
        cid = torch.empty(m, dtype=torch.int32, device=dev)
        w = torch.empty(m, dtype=torch.float32, device=dev)
        xq = torch.empty((m, 3), dtype=torch.float32, device=dev)
        if m:
            call('nksr_chunk_pair_fill', C.byref(self._cgrid), 1, ptr(xyz), n, ptr(self._chunk_flag), ptr(offs), ptr(q), ptr(cid), ptr(w),
                 ptr(xq), stream())
        return offs, q, cid, w, xq

    def _evaluate_f_model(self, xyz, grad, max_points=1 << 22):
        n = xyz.shape[0]
        xyz = xyz.contiguous()
        f_out = torch.zeros(n, dtype=torch.float32, device=xyz.device)
        g_out = torch.zeros((n, 3), dtype=torch.float32, device=xyz.device) if grad else None
        if n == 0:
            return EvaluationResult(f_out, g_out)
        offs, _, cid, w, xq = self._pairs(xyz)
        m = xq.shape[0]
        if m == 0:                                 # no chunk weighs at any query: f = 0
            return EvaluationResult(f_out, g_out)
        # (a part parked on chunk_tmp_device / by to_('cpu') is borrowed for its evaluation: out-of-core, one part resident at a time)
        if len(self.parts) == 1:                   # one evaluation call per part for ALL pairs
            with borrowed(self.parts[0].field, self.home) as pf:
                res = pf._evaluate_f_model(xq, grad, max_points)
            f, gr = res.value.contiguous(), (res.gradient.contiguous() if grad else None)
        else:
            f = torch.empty(m, dtype=torch.float32, device=xyz.device)
            gr = torch.empty((m, 3), dtype=torch.float32, device=xyz.device) if grad else None
            pl = self._part_lut[cid.long()]
            for pi, part in enumerate(self.parts):
                s = torch.nonzero(pl == pi).reshape(-1)
                if s.numel():
                    with borrowed(part.field, self.home) as pf:
                        res = pf._evaluate_f_model(xq[s].contiguous(), grad, max_points)
                    f[s] = res.value
                    if grad:
                        gr[s] = res.gradient
        # f = sum w f_c / max(sum w, 1e-20); the gradient of the weights is ignored (they are flat outside the seams)
        call('nksr_chunk_blend', n, ptr(offs), ptr(w), ptr(f), ptr(gr) if grad else None, ptr(f_out), ptr(g_out) if grad else None, stream())
        return EvaluationResult(f_out, g_out)

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
        return self.owns_points((ijk.to(torch.float32) + 0.5) * self.svh.voxel_size)

    def base_cell_halo_mask(self, ijk):
        """Owned cells plus one ring of neighbours: the MISE hanging-vertex rule needs to know whether
        the cells across a rank seam were refined, so they are evaluated (but not meshed) here too."""
        if self.world_size == 1:
            return torch.ones(ijk.shape[0], dtype=torch.bool, device=ijk.device)
        w = self.svh.voxel_size
        return self.near_owned((ijk.to(torch.float32) + 0.5) * w, w)

    def seam_flags(self, edge_vkey, edge_axis, cells_per_voxel):
        """uint8 per mesh vertex of this rank's piece: 1 when another rank may emit the vertex too -- one of the four lattice cells
        around its edge is not this rank's (csrc/chunks.hip k_edge_seam_flags: base_cell_mask's arithmetic).  Rank 0 then groups only
        those (dist.merge_meshes)."""
        n = int(edge_vkey.numel())
        flags = torch.empty(n, dtype=torch.uint8, device=edge_vkey.device)
        if n:
            if getattr(self, '_owner_dev', None) is None:
                self._owner_dev = torch.tensor(self.owner, dtype=torch.int32, device=edge_vkey.device)
            call('nksr_edge_seam_flags', C.byref(self._cgrid), ptr(edge_vkey.contiguous()), ptr(edge_axis.to(torch.int8).contiguous()), n, int(cells_per_voxel),
                 float(np.float32(self.svh.voxel_size)), ptr(self._owner_dev), int(self.rank), ptr(flags), stream())
        return flags

    def _owner_flags(self, xyz, reach):
        """csrc/chunks.hip k_points_owner_flags: one launch instead of the ~100 torch launches of _near_owned_torch (same arithmetic)."""
        xyz = xyz.to(torch.float32).contiguous()
        n = xyz.shape[0]
        flags = torch.empty(n, dtype=torch.uint8, device=xyz.device)
        if n:
            if getattr(self, '_owner_dev', None) is None:
                self._owner_dev = torch.tensor(self.owner, dtype=torch.int32, device=xyz.device)
            call('nksr_points_owner_flags', C.byref(self._cgrid), ptr(xyz), n, float(np.float32(reach)), ptr(self._owner_dev), int(self.rank), ptr(flags), stream())
        return flags.bool()

    def owns_points(self, xyz):
        """Points (model units) inside a core this rank owns."""
        if self.world_size == 1:
            return torch.ones(xyz.shape[0], dtype=torch.bool, device=xyz.device)
        if xyz.is_cuda:
            return self._owner_flags(xyz, 0.0)
        own = torch.tensor(self.owner, dtype=torch.long, device=xyz.device)
        return own[self.chunk_of(xyz)] == self.rank

    def near_owned(self, centers, reach):
        """Points whose box centre +- ``reach`` (along the split axes) touches a core this rank owns."""
        if self.world_size == 1:
            return torch.ones(centers.shape[0], dtype=torch.bool, device=centers.device)
        if centers.is_cuda:
            return self._owner_flags(centers, reach)
        return self._near_owned_torch(centers, reach)

    def _near_owned_torch(self, centers, reach):
        """near_owned in torch operations (the specification of k_points_owner_flags; tests compare the two)."""
        dev = centers.device
        own = torch.tensor(self.owner, dtype=torch.long, device=dev)
        w = reach
        # chunk index of centre - w / centre / centre + w along every split axis; only points next to a chunk
        # boundary (lo != hi on some axis) can see a different owner than their own chunk's
        split = [a for a in range(3) if self.grid[a] > 1]
        idx = {}
        for a in split:
            for k, off in ((0, -w), (1, 0.0), (2, w)):
                idx[(a, k)] = torch.floor((centers[:, a] + off - self.origin[a]) / self.chunk_size).long().clamp_(0, self.grid[a] - 1)

        def lin(sel, ks):
            out = torch.zeros(1, dtype=torch.long, device=dev)
            for a in range(3):
                ia = idx[(a, ks[a])][sel] if a in split else 0
                out = out * self.grid[a] + ia
            return out

        every = slice(None)
        m = own[lin(every, (1, 1, 1))] == self.rank
        near = torch.zeros(centers.shape[0], dtype=torch.bool, device=dev)
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
        if self.world_size == 1 and not self.distributed:
            return res
        import time
        on_gpu = res.v.is_cuda and torch.cuda.is_available()      # (the gather also runs under gloo with CPU tensors: tests/test_dist_cpu.py)
        if on_gpu:
            torch.cuda.current_stream().synchronize()
        t0 = time.perf_counter()
        v, f = D.gather_meshes(res.v, res.f, res.edge_vkey, res.edge_axis, seam=getattr(res, 'seam_flag', None))
        if on_gpu:
            torch.cuda.current_stream().synchronize()
        self.last_gather_s = time.perf_counter() - t0      # mesh gather (point-to-point to rank 0) + seam merge
        res.v, res.f = v, f
        res.c = self.texture_field.evaluate_color(v) if self.texture_field is not None else None
        return res

    def finalize_mesh_named(self, res):
        """The adaptive dual graph's pieces: vertices named by the ordered pair of primal cells they join -- (size, key) names, the
        same on every rank -- gathered on rank 0 and merged there (dist.merge_named)."""
        if self.world_size == 1 and not self.distributed:
            return res
        import time
        on_gpu = res.v.is_cuda and torch.cuda.is_available()
        if on_gpu:
            torch.cuda.current_stream().synchronize()
        t0 = time.perf_counter()
        v, f, names = D.gather_named(res.v, res.f, res.vertex_names5)
        if on_gpu:
            torch.cuda.current_stream().synchronize()
        self.last_gather_s = time.perf_counter() - t0
        res.v, res.f, res.vertex_names5 = v, f, names
        res.c = self.texture_field.evaluate_color(v) if self.texture_field is not None else None
        return res

    def for_rank(self, rank, world_size, fields):
        """Same scene seen from another (simulated) rank holding the per-chunk ``fields`` -- test helper."""
        parts = [ChunkPart(f, [c], self.frame, solved=bool(f.solve_info)) for c, f in sorted(fields.items())]
        return MultiChunkField(parts, self.cores, self.ov, self.origin, self.chunk_size, self.grid, self.owner, rank,
                               world_size, self.frame, self.interpolators, self.svh.device, adaptive_depth=self.meshing_depth,
                               halo_inner=self.halo_inner)

    def to_(self, device):
        """``to_('cpu')`` parks the parts and the union grid on the host (NKSR-USAGE.md:163: "Put everything onto CPU"); evaluation
        and ``extract_dual_mesh`` keep running on the GPU the field was made on, borrowing one part at a time (out-of-core meshing:
        peak HBM = one part + the union grid + the lattice, not the scene).  The chunk tables (a few KB) stay where they are."""
        for p in self.parts:
            p.field.to_(device)
        self.svh.to_(device)
        return self

    def evaluate_f(self, xyz, grad=False):
        return super().evaluate_f(xyz.to(self.home), grad)

    @torch.no_grad()
    def extract_dual_mesh(self, mise_iter=0, grid_upsample=1, max_points=-1):
        with borrowed(self.svh, self.home):
            return super().extract_dual_mesh(mise_iter=mise_iter, grid_upsample=grid_upsample, max_points=max_points)


# ---- the batched solve ------------------------------------------------------------------------------------------------------
def select_chunk_points(xyz, lo, grid, chunk_size, band, wanted):
    """(point index, chunk id) of every point inside core +- band of a chunk in ``wanted`` (bool per chunk), sorted by
    (chunk, point index), + the points per chunk.  Same comparisons as a per-chunk boolean mask: x >= lo_c - band and
    x < hi_c + band along the split axes, the bounds rounded to fp32."""
    dev = xyz.device
    n = xyz.shape[0]
    nchunk = grid[0] * grid[1] * grid[2]
    z = torch.zeros(0, dtype=torch.long, device=dev)
    if n == 0:
        return z, z, [0] * nchunk
    xyz = xyz.contiguous()
    G, keep = chunk_grid_struct(lo, grid, chunk_size, band, None, dev)
    flag = torch.tensor([0 if w else -1 for w in wanted], dtype=torch.int32, device=dev)
    cnt = torch.empty(n + 1, dtype=torch.int32, device=dev)
    cnt[n:] = 0
    call('nksr_chunk_pair_counts', C.byref(G), 0, ptr(xyz), n, ptr(flag), ptr(cnt), stream())
    offs = ops.exclusive_sum_i32(cnt)
    m = int(offs[n].item())
    if m == 0:
        return z, z, [0] * nchunk
    idx = torch.empty(m, dtype=torch.int64, device=dev)
    cid = torch.empty(m, dtype=torch.int32, device=dev)
    call('nksr_chunk_pair_fill', C.byref(G), 0, ptr(xyz), n, ptr(flag), ptr(offs), ptr(idx), ptr(cid), None, None, stream())
    cid = cid.long()
    if nchunk > 1:
        # the fill order is (point, chunk): a STABLE sort on the chunk bits alone gives (chunk, point) -- one radix pass over 6-8 bits
        ks, order = ops.sort_pairs(cid, torch.arange(m, dtype=torch.int32, device=dev), end_bit=ops._bits(nchunk))
        order = order.long()
        idx, cid = idx[order], ks
    counts = torch.bincount(cid, minlength=nchunk).tolist()
    return idx, cid, counts


def reconstruct_by_chunk(rec, xyz, normal, sensor, chunk_size, overlap_ratio, approx_kernel_grad, solver_max_iter,
                         solver_tol, fused_mode, preprocess_fn, sim=None, sharded_input=False, chunk_owner=None, chunk_bounds=None, sim_exchange=None):
    """``sim=(rank, world_size)`` runs one simulated rank without a process group (tests); with ``sim_exchange(local, dest_of) ->
    payload`` the simulated rank goes through the halo exchange step too, the callable standing in for the collectives
    (tools/prof_rank_tail.py: everything a rank of N does after its solve, timed on one GPU).
    ``sharded_input``: every rank passes only ITS part of the cloud -- at least the points inside core +- band of the
    chunks it owns ("each rank receives only its chunks' points (+overlap)").  The chunk grid then
    comes from ``chunk_bounds`` = (lo[3], hi[3]) or from an all_reduce of the local bounding boxes, the per-core point
    counts from an all_reduce(MAX) (some rank holds every core completely), and ``chunk_owner`` (list, one rank per
    chunk) lets the caller that distributed the data dictate the ownership it assumed."""
    hp = rec.hparams
    dev = rec.device
    rank, ws = sim if sim is not None else D.world()
    active = D.active() and sim is None            # a process group takes part (world > 1, or forced at world 1: NKSR_DIST_FORCE)
    collective = sharded_input and active
    from .density import bbox_center
    if xyz.shape[0]:
        lo_t, hi_t, _ = bbox_center(xyz)
        if not bool(torch.isfinite(torch.cat([lo_t.reshape(-1), hi_t.reshape(-1)])).all()):   # (nksr_bbox: NaN in lo[0] when any coordinate is NaN / infinite; the CPU branch: the extrema themselves)
            raise RuntimeError('non-finite coordinates in the input')
    if chunk_bounds is not None:
        lo, hi = [float(v) for v in chunk_bounds[0]], [float(v) for v in chunk_bounds[1]]
    else:
        if xyz.shape[0]:
            pass
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
    frame = ChunkFrame(hp.voxel_size, hp.tree_depth, lo, grid, chunk_size, band)
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
    jobs = [c for c in range(nchunk) if owner[c] == rank and counts[c] > 0]
    for c in jobs:
        if sharded_input and local_counts[c] != counts[c]:
            raise RuntimeError('sharded input: rank %d owns chunk %d but holds %d of its %d core points' % (rank, c, local_counts[c], counts[c]))
    wanted = [False] * nchunk
    for c in jobs:
        wanted[c] = True
    pidx, pcid, npts = select_chunk_points(xyz, lo, grid, chunk_size, band, wanted)
    bx, bn, bc = xyz[pidx], (normal[pidx] if normal is not None else None), pcid
    if preprocess_fn is not None:
        # the reference's contract: preprocess_fn sees the points of ONE chunk, in the caller's coordinates, on the calling thread
        xs_, ns_, cs_ = [], [], []
        bs = sensor[pidx] if sensor is not None else None
        o = 0
        for c in range(nchunk):
            m = npts[c]
            if m == 0:
                continue
            sl = slice(o, o + m)
            o += m
            try:
                cx_, cn_, _ = preprocess_fn(bx[sl].contiguous(), bn[sl].contiguous() if bn is not None else None,
                                            bs[sl].contiguous() if bs is not None else None)
            except ChunkTooSmall:
                npts[c] = 0
                continue
            if cn_ is None:
                raise RuntimeError('oriented input required (normal= or sensor= with a normal-estimating preprocess_fn)')
            npts[c] = int(cx_.shape[0])
            xs_.append(cx_)
            ns_.append(cn_.to(torch.float32))
            cs_.append(torch.full((cx_.shape[0],), c, dtype=torch.long, device=dev))
        bx = torch.cat(xs_) if xs_ else xyz[:0]
        bn = torch.cat(ns_) if ns_ else xyz[:0]
        bc = torch.cat(cs_) if cs_ else pcid[:0]
    if bn is None:
        raise RuntimeError('oriented input required (normal= or sensor= with a normal-estimating preprocess_fn)')
    small = [c for c in jobs if npts[c] < MIN_CHUNK_POINTS]          # a handful of stray points: nothing to solve, neighbours cover the band
    if small:
        keep = torch.ones(nchunk, dtype=torch.bool, device=dev)
        keep[small] = False
        s = torch.nonzero(keep[bc]).reshape(-1)
        bx, bn, bc = bx[s], bn[s], bc[s]
        for c in small:
            npts[c] = 0
    jobs = [c for c in jobs if npts[c] >= MIN_CHUNK_POINTS]
    # (non-finite normals: caught by the box readback every batch starts with, Reconstructor._key_bits)
    # sub-batches of whole chunks (memory: ~2 KB per point at tree_depth 5), chunks of a batch in slot order
    jobs.sort(key=lambda c: frame.key_range(c)[0])
    budget = int(getattr(rec, 'chunk_batch_points', 0) or 0)
    if budget <= 0:
        # automatic: as many chunks per solve as the FREE memory of the device holds (the reference's chunk mode exists to bound
        # memory, examples/recons_by_chunk.py:17-18) -- at most 2^25 points.  A batched solve takes ~4.5 KB of HBM per solved point
        # at tree_depth 5 (84.7 GB for the 19.7 M band-included points of the 64-chunk scene; kernel rows are 45 % of it), in
        # proportion to the depth; 70 % of what is free (+ what torch's allocator holds unused) may be planned with.  Results do not
        # depend on the split (tests/test_gpu_full_size.py: a chunk alone == the chunk in the batch, bit for bit).
        budget = 1 << 25
        if dev.type == 'cuda' and fused_mode:
            free = float(os.environ.get('NKSR_FREE_HBM_GB', 0)) * 1e9
            if free <= 0:
                free = torch.cuda.mem_get_info(dev)[0] + torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)
            per_point = 4500.0 * hp.tree_depth / 5.0 * (0.9 if str(getattr(rec, 'row_format', None) or os.environ.get('NKSR_ROW_FORMAT')) == 'factors' else 1.0)      # (factor records are a fifth of the rows, but the set-up sweep holds the dense rows of the coarse levels beside them: peak ~0.9)
            budget = int(max(min(budget, 0.7 * free / per_point), 1))
    if not fused_mode:
        budget = 0        # the assembled solve (fused_mode=False) has no segmented form: one chunk per solve, as the reference runs them
    batches, cur, acc = [], [], 0
    for c in jobs:
        if cur and acc + npts[c] > budget:
            batches.append(cur)
            cur, acc = [], 0
        cur.append(c)
        acc += npts[c]
    if cur:
        batches.append(cur)
    shift_all = torch.from_numpy(np.stack([frame.shift(c) for c in range(nchunk)])).to(dev)
    slot_org = torch.from_numpy(np.stack([np.asarray(frame.slot_origin(c), np.float64) * frame.w0 for c in range(nchunk)]).astype(np.float32)).to(dev)
    parts, timing = [], {}
    for ids in batches:
        if len(batches) > 1:
            inb = torch.zeros(nchunk, dtype=torch.bool, device=dev)
            inb[ids] = True
            s = torch.nonzero(inb[bc]).reshape(-1)
            x_b, n_b, c_b = bx[s], bn[s], bc[s]
        else:
            x_b, n_b, c_b = bx, bn, bc
        xs = (x_b + shift_all[c_b]).contiguous()                      # the exploded frame: x' = x + T_c (one fp32 rounding)
        rlo, rhi, _ = bbox_center((xs - slot_org[c_b]).contiguous())
        rlo, rhi = rlo.tolist(), rhi.tolist()
        if min(rlo) < frame.usable_lo or max(rhi) >= frame.usable_hi:
            raise RuntimeError('chunk data leaves its slot of the exploded frame (extent %s .. %s, usable %.3f .. %.3f): points outside '
                               'chunk_bounds along an axis that is not split?' % (rlo, rhi, frame.usable_lo, frame.usable_hi))
        kr = [frame.key_range(c) for c in ids]
        fld = rec._reconstruct_single(xs, n_b.to(torch.float32).contiguous(), approx_kernel_grad, solver_max_iter, solver_tol, fused_mode,
                                      chunks=(ids, [k[0] for k in kr], [k[1] for k in kr], frame) if fused_mode else None)
        fld.matrix = None
        fld._fused_op = None
        for k, v in getattr(fld, 'timing', {}).items():
            timing[k] = timing.get(k, 0.0) + v
        if rec.chunk_tmp_device != dev and not active and sim is None and len(batches) > 1:
            fld.to_(rec.chunk_tmp_device)  # reference semantics: park solved chunks elsewhere (only useful when there are several batches)
            if getattr(rec, 'chunk_spill_dir', None) and torch.device(rec.chunk_tmp_device).type == 'cpu':
                timing['spilled_bytes'] = timing.get('spilled_bytes', 0) + spill_to_disk(fld, rec.chunk_spill_dir)
        parts.append(ChunkPart(fld, ids, frame))
    rec.timing = timing
    interps = rec.network.interpolators
    t_x = _now(rec)
    inner = halo_inner(hp.voxel_size, hp.adaptive_depth, getattr(rec, 'dual_graph', 'lattice'))      # how deep a halo reaches into its own core
    if active or (sim is not None and sim_exchange is not None):
        # the exchange carries the halo of every chunk (the voxels other ranks can touch), not the whole field; which
        # chunks were actually solved travels with it (a sparse chunk may have been skipped by its owner)
        def band_of(c):
            return exchange_band(cores[c], frame.chunk3(c), grid, ov, hp.voxel_size, inner)
        local = {}
        for p in parts:
            local.update(p.pack_halos({c: band_of(c) for c in p.ids}))
        # a rank only evaluates the blend inside its own cores (+ the halo ring it evaluates): it needs exactly the chunks whose
        # weight support (core +- ov) reaches there -- its spatial neighbours, not all N.  Who needs what is geometry (cores, owners,
        # which cores hold points): every rank computes the same table, so a halo is SENT only to the ranks that need it
        # (all_to_all with per-pair sizes; a chunk its owner skipped is simply not sent)
        dest_of = halo_destinations(cores, ov + inner, grid, owner, counts, ws)
        payload = D.exchange_payloads_to(local, dest_of) if active else sim_exchange(local, dest_of)
        mine = set(local)
        need = sorted(c for c in payload if c not in mine)
        if need:
            remote = fields_from_payloads([(frame.key_range(c)[0], payload[c][0], payload[c][1]) for c in need], hp.voxel_size, interps, dev)
            remote.meshing_depth = int(hp.adaptive_depth)
            if remote.mask_field is None:
                remote.set_mask_field(LayerField(remote.svh, hp.adaptive_depth))
            parts.append(ChunkPart(remote, sorted(need, key=lambda c: frame.key_range(c)[0]), frame, solved=False))
        timing['t_exchange'] = _now(rec) - t_x          # halo exchange: pack, size + byte collectives, the remote field's tables
    # (batches parked on chunk_tmp_device stay there: the blend borrows one part at a time -- borrowed() -- so meshing a scene
    # whose chunks do not fit the GPU together works as the reference's small-memory recipe says, NKSR-USAGE.md:150-167)
    mf = MultiChunkField(parts, cores, ov, lo, chunk_size, grid, owner, rank, ws, frame, interps, dev, distributed=active,
                         adaptive_depth=int(hp.adaptive_depth), halo_inner=inner)
    mf.dual_graph = getattr(rec, 'dual_graph', 'lattice')
    return mf


def _now(rec):
    import time
    if getattr(rec, 'sync_timing', False):
        torch.cuda.current_stream().synchronize()
    return time.perf_counter()


def halo_destinations(cores, margin, grid, owner, counts, world_size):
    """{chunk: [ranks that need its halo and do not own it]} -- pure geometry (cores, owners, which chunks hold points), so every
    rank computes the same table and a halo is sent only where it is read."""
    nonempty = [c for c in range(len(owner)) if counts[c] > 0]
    dest_of = {}
    for r in range(world_size):
        owned_r = [c for c in nonempty if owner[c] == r]
        for c in needed_chunks(cores, margin, grid, owned_r, nonempty):
            if owner[c] != r:
                dest_of.setdefault(c, []).append(r)
    return dest_of


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
