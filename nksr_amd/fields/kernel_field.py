"""Neural kernel field -- host-side mirror of ``nksr.fields.KernelField``.

Reference interface (call sites): ``KernelField(svh=, interpolator=, features=,
approx_kernel_grad=)`` models/nksr_net.py:91-96; ``solver_config['verbose']`` :97-98;
``solve_non_fused(pos_xyz=, normal_xyz=, normal_value=, pos_weight=, normal_weight=,
reg_weight=)`` :105-112; fused ``solve`` selected by ``fused_mode`` examples/recons_waymo.py:33;
``evaluate_f`` / ``evaluate_f_bar`` models/loss.py:99,189-198.

Math: unknowns alpha (one per voxel per level, levels
concatenated fine -> coarse), f(x) = sum_d sum_{j in N27(x)} alpha_j <phi_d(x), psi_j> B(.),
normal equations (w_p G^T G + w_n Q^T Q + reg I) alpha = w_n Q^T n solved by Jacobi-PCG.
All numeric work runs in HIP kernels (csrc/kfield.hip, assemble.hip, pcg.hip).
"""
import ctypes as C
import os
import time

import torch

from .. import _lib, ops
from .._lib import PC_MAX_STEPS, CoarsePrecondT, FusedOpT, HierT, SegmentsT, SiteSetT, call, ptr, stream
from .base_field import BaseField, EvaluationResult


def pack_interpolator(interp):
    """Flat fp32 weights W1[H,K] b1[H] W2[H,H] b2[H] W3[K,H] b3[K] of one level."""
    return interp.packed()


PC_RATIO = 40.0          # Chebyshev interval [lambda_max / PC_RATIO, lambda_max] of the coarse block (100 until late round 3: 11.16 -> 10.9
#                          PCG iterations per chunk of the 64-chunk scene at the same step count; 10..20 are worse again, 200 much worse)
SMALL_FIELD_UNKNOWNS = 1 << 16      # single fields up to this size take the coarse-level block at once, from level 1 (solve_fused)
SMALL_FIELD_CHECK_EVERY = 6
SMALL_FIELD_PC = {'first_level': 1, 'steps': 8, 'ratio': 40.0}
PC_DROP_TOL = 0.005        # packed coarse block: off-diagonal entries below this fraction of the (unit) diagonal are left out
_DETAIL = os.environ.get('NKSR_TIMING_DETAIL', '') == '1'
DETAIL_TIMES = {}


def _tick(name, t0):
    """NKSR_TIMING_DETAIL=1: synchronised sub-stage times accumulated in DETAIL_TIMES (diagnostics only)."""
    if not _DETAIL:
        return t0
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    DETAIL_TIMES[name] = DETAIL_TIMES.get(name, 0.0) + (t1 - t0)
    return t1


class Segments:
    """Independent diagonal blocks of ONE hierarchy (nksr_segments_t): the chunks of a batched chunk solve.  Segment i owns
    the Morton key range [key_lo[i], key_hi[i]) of the finest level (an aligned cube of the lattice: its ancestors' ranges are the
    shifted ones), hence one contiguous run of voxels per level and one contiguous run of the Morton-sorted sites.  Everything is
    derived on the device (searchsorted), no host sync."""

    def __init__(self, svh, key_lo, key_hi, ids=None):
        dev = svh.device
        self.key_lo, self.key_hi = key_lo.to(dev, torch.int64).contiguous(), key_hi.to(dev, torch.int64).contiguous()
        self.nseg, self.nranges = int(self.key_lo.numel()), svh.depth
        self.ids = list(ids) if ids is not None else list(range(self.nseg))
        off = svh.offsets
        lo, hi, segs = [], [], []
        for d in range(svh.depth):
            k = svh.level(d).keys
            a = torch.searchsorted(k, self.key_lo >> (3 * d))
            b = torch.searchsorted(k, self.key_hi >> (3 * d))
            lo.append(a + off[d])
            hi.append(b + off[d])
            segs.append(torch.bucketize(k, self.key_lo >> (3 * d), right=True) - 1)
        self.lo = torch.stack(lo, 1).to(torch.int32).contiguous()          # [nseg, L]
        self.hi = torch.stack(hi, 1).to(torch.int32).contiguous()
        self.unknown_seg = torch.cat(segs).to(torch.int32).contiguous()     # [M]
        self.info = torch.zeros((self.nseg, 2), dtype=torch.float64, device=dev)
        self.c = SegmentsT()
        self.c.nseg, self.c.nranges = self.nseg, self.nranges
        self.c.lo, self.c.hi, self.c.info = ptr(self.lo), ptr(self.hi), ptr(self.info)

    def of_keys(self, keys0):
        """segment of level-0 Morton keys"""
        return torch.bucketize(keys0, self.key_lo, right=True) - 1


class KernelField(BaseField):
    def __init__(self, svh, interpolator, features, approx_kernel_grad=False):
        super().__init__(svh)
        self.approx_kernel_grad = bool(approx_kernel_grad)
        self.solver_config = {'verbose': False, 'max_iter': 2000, 'tol': 1e-5, 'check_every': 16, 'sync_timing': False}
        self.solve_info = {}
        self.kdim = int(interpolator[0].kernel_dim)
        self.hidden = int(interpolator[0].hidden_dim)
        dev = svh.device
        self._mlp = [pack_interpolator(interpolator[d]).detach().to(dev, torch.float32).contiguous() for d in range(svh.depth)]
        self._interps_in, self._feat_in = list(interpolator), list(features)      # the caller's tensors / modules (autograd, training path)
        self._feat, self._psi = [], []
        for d in range(svh.depth):
            n = svh.num_voxels(d)
            f = features[d] if features[d] is not None else torch.zeros((0, self.kdim), device=dev)
            f = f.detach().to(dev, torch.float32).contiguous()
            if f.shape != (n, self.kdim):
                raise RuntimeError('basis_features[%d] has shape %s, expected (%d, %d)' % (d, tuple(f.shape), n, self.kdim))
            psi = torch.empty_like(f)
            call('nksr_voxel_psi', ptr(f), n, self.kdim, self.hidden, ptr(self._mlp[d]), ptr(psi), stream())
            self._feat.append(f)
            self._psi.append(psi)
        self._apsi_key = None
        self.alpha = torch.zeros(svh.num_unknowns, dtype=torch.float32, device=dev)
        self._hier = self._make_hier()
        self.matrix = None  # (rowptr, cols, vals, diag) of the last non-fused solve

    # alpha is written by HIP kernels through raw pointers (no torch version bump) and the caching allocator recycles
    # addresses, so the evaluation cache (_alpha_hier) is keyed on ASSIGNMENT: every ``fld.alpha = ...`` drops it
    @property
    def alpha(self):
        return self._alpha

    @alpha.setter
    def alpha(self, value):
        self._alpha = value
        self._apsi_key = None
        self._apsi = self._apsi_hier = None

    def invalidate_alpha_cache(self):
        """Call after writing into ``alpha``'s storage in place (kernels, ``alpha[...] = ``)."""
        self._apsi_key = None
        self._apsi = self._apsi_hier = None

    # ---- C struct describing the hierarchy + features ------------------------------------------
    def _make_hier(self):
        h = HierT()
        svh = self.svh
        h.depth, h.kdim, h.hidden, h.inv_w0 = svh.depth, self.kdim, self.hidden, svh.inv_w0
        off = svh.offsets
        for d in range(svh.depth):
            g = svh.level(d)
            lv = h.lv[d]
            lv.n, lv.offset = g.num_voxels, off[d]
            lv.keys, lv.ijk, lv.nbr = ptr(g.keys), ptr(g.ijk), ptr(g.nbr)
            lv.hkeys, lv.hvals, lv.hcap = ptr(g.hash.hkeys), ptr(g.hash.hvals), g.hash.cap
            lv.feat, lv.psi, lv.mlp = ptr(self._feat[d]), ptr(self._psi[d]), ptr(self._mlp[d])
        return h

    # ---- kernel rows ---------------------------------------------------------------------------------
    def kernel_rows(self, xyz, grad, scale=1.0, values=True, hier=None):
        """Dense-slot rows: val [n, L, 27] (``None`` with values=False) and (grad) dval [n, 3, L, 27]
        (model units), times ``scale``.  ``hier``: a masked copy of the hierarchy (_coarse_hier): rows of masked levels are 0."""
        n, L = xyz.shape[0], self.svh.depth
        val = torch.empty((n, L, 27), dtype=torch.float32, device=self.device) if values else None
        dval = torch.empty((n, 3, L, 27), dtype=torch.float32, device=self.device) if grad else None
        call('nksr_kernel_rows', C.byref(hier if hier is not None else self._hier), ptr(xyz), n, int(self.approx_kernel_grad), float(scale), None, 0, None, None,
             ptr(val), ptr(dval), stream())
        return val, dval

    def _coarse_hier(self, c0):
        """The hierarchy restricted to its levels >= c0: the finer levels are EMPTY (no voxels, no hash: every site misses them),
        the coarse ones keep their level index -- and with it their geometry -- and are re-based to unknown 0."""
        h = HierT.from_buffer_copy(self._hier)
        off = self.svh.offsets
        for d in range(self.svh.depth):
            if d < c0:
                h.lv[d].n, h.lv[d].offset, h.lv[d].hcap = 0, 0, 0
            else:
                h.lv[d].offset = off[d] - off[c0]
        return h

    def kernel_rows_level_major(self, xyz, grad, scale, out, level_stride, row_index=None, row_cells=None, site_scale=None):
        """Rows of the sites ``xyz`` written LEVEL-MAJOR into ``out`` ([L, level_stride, 27]): position rows (grad=False, one per
        site) or gradient rows (three per site), site i at row ``row_index[i]`` (default i * rows-per-site); ``row_cells``
        [L, level_stride] receives the level-d cell (global unknown index) of every row written."""
        call('nksr_kernel_rows', C.byref(self._hier), ptr(xyz), xyz.shape[0], int(self.approx_kernel_grad), float(scale), ptr(site_scale), int(level_stride),
             ptr(row_index), ptr(row_cells), None if grad else ptr(out), ptr(out) if grad else None, stream())

    def kernel_factors_level_major(self, xyz, grad, scale, vec, pos, level_stride, row_index=None, row_cells=None, site_scale=None):
        """The rank-4 factor records of the sites' rows (csrc/kfield.hip: k_kernel_factors; kernel_dim 4): ``vec`` [L, level_stride, 4],
        ``pos`` [level_stride, 4]; a position site owns one row, a normal site (grad=True) four (header + one per axis)."""
        call('nksr_kernel_factors', C.byref(self._hier), ptr(xyz), xyz.shape[0], int(bool(grad)), int(self.approx_kernel_grad), float(scale),
             ptr(site_scale), int(level_stride), ptr(row_index), ptr(row_cells), ptr(vec), ptr(pos), stream())

    def _row_format(self):
        """'dense' (the default: 108 bytes per row and level, the sweep streams them -- HBM-bound) or 'factors' (kernel_dim 4 only,
        opt-in by solver_config['row_format'] / NKSR_ROW_FORMAT: 16-byte records per row and level, the sweep rebuilds the 27
        slots in registers -- a fifth of the memory (8 GB instead of 38 GB on the 64-chunk scene), but bound by vector-ALU issue:
        22 ms per application there against 11 ms)."""
        want = self.solver_config.get('row_format') or os.environ.get('NKSR_ROW_FORMAT') or 'auto'
        if want not in ('auto', 'factors', 'dense'):
            raise RuntimeError("row_format must be 'auto', 'factors' or 'dense'")
        if want == 'factors' and self.kdim != 4:
            raise RuntimeError('the factor form of the kernel rows needs kernel_dim 4')
        return 'factors' if (want == 'factors' and self.kdim == 4) else 'dense'

    def _sorted_sites(self, xyz):
        """Permutation that Morton-sorts sites by their level-0 cell + the sorted keys."""
        n = xyz.shape[0]
        keys = torch.empty(n, dtype=torch.int64, device=self.device)
        call('nksr_point_keys', ptr(xyz), n, self.svh.inv_w0, ptr(keys), stream())
        idx = torch.arange(n, dtype=torch.int32, device=self.device)
        ks, perm = ops.sort_pairs(keys, idx, level=0)
        return ks, perm.long()

    def _site_ranges(self, site_keys):
        starts, ends = [], []
        for d in range(self.svh.depth):
            g = self.svh.level(d)
            s = torch.empty(g.num_voxels, dtype=torch.int32, device=self.device)
            e = torch.empty(g.num_voxels, dtype=torch.int32, device=self.device)
            call('nksr_site_ranges', ptr(site_keys), site_keys.numel(), ptr(g.keys), g.num_voxels, d, ptr(s), ptr(e), stream())
            starts.append(s)
            ends.append(e)
        return starts, ends

    # ---- assembly -----------------------------------------------------------------------------------
    def assemble(self, pos_xyz, normal_xyz, normal_value, pos_weight, normal_weight, reg_weight=1.0,
                 pos_sorted_keys=None, normal_sorted_keys=None, coarse_from=None, fused_op=None):
        """Materialise the CSR normal equations.  Returns (rowptr, cols, vals, diag, b).
        ``*_sorted_keys``: level-0 Morton keys of site sets that are ALREADY Morton-sorted.
        ``coarse_from`` = c0: only the diagonal block of the levels >= c0 (plain CSR, local indices) -- the preconditioner's;
        with ``fused_op`` (fused_operator's result) it reads the kernel rows the matrix-free operator already holds."""
        dev = self.device
        hier = self._hier if coarse_from is None else self._coarse_hier(int(coarse_from))
        M = self.svh.num_unknowns if coarse_from is None else self.svh.num_unknowns - self.svh.offsets[int(coarse_from)]
        if M == 0:
            raise RuntimeError('empty hierarchy')
        keep = []  # keep every buffer alive until the launches are enqueued
        sets = (SiteSetT * 2)()
        nsets = 0
        if fused_op is not None:
            # the operator's Morton-ordered row list IS a site set with one row per "site": the rows of a cell are the run
            # span[0][j] .. span[1][j] (fused tables), so neither site ranges nor a row index are needed
            first, last = fused_op['span'][0], fused_op['span'][1]
            st_all, en_all = first.clamp(min=0), (last + 1).contiguous()
            off = self.svh.offsets
            S = sets[0]
            S.n, S.ncomp, S.weight = fused_op['rows_total'], 1, 1.0
            if fused_op.get('row_format') == 'factors':
                # the factor form holds no dense rows: those of the levels >= coarse_from were written out by the set-up sweep
                # (or are expanded now); the array starts at level coarse_from (nksr_siteset_t.level_base)
                dense = self._dense_coarse_rows(fused_op, int(coarse_from))
                S.val, S.level_base = ptr(dense), int(coarse_from)
                keep.append(dense)
            else:
                S.val = ptr(fused_op['rows_all'])
                if fused_op.get('compact'):      # the compact array: the assembly finds a cell's rows through the operator's tables
                    S.compact_cells, S.compact_nbr32 = ptr(fused_op['row_cells']), ptr(fused_op['nbr32'])
            S.level_stride = fused_op['rows_total']
            for d in range(self.svh.depth):
                nd = self.svh.level(d).num_voxels
                S.start[d], S.end[d] = ptr(st_all[off[d]:off[d] + nd]), ptr(en_all[off[d]:off[d] + nd])
            keep += [st_all, en_all]
            nsets = 1
        for xyz, target, weight, ncomp, pre in (((pos_xyz, None, pos_weight, 1, pos_sorted_keys),
                                                  (normal_xyz, normal_value, normal_weight, 3, normal_sorted_keys)) if fused_op is None else ()):
            if xyz is None or xyz.shape[0] == 0:
                continue
            xyz = xyz.to(dev, torch.float32).contiguous()
            if pre is not None:
                ks, perm, xs = pre, None, xyz
            else:
                ks, perm = self._sorted_sites(xyz)
                xs = xyz[perm].contiguous()
            if not float(weight) >= 0.0:
                raise RuntimeError('solver weights must be >= 0')
            # rows (and targets) are produced pre-multiplied by sqrt(weight): the Gram products of the
            # assembly are then bitwise symmetric and its matrix-core operands need no scaling
            sw = float(weight) ** 0.5
            val, dval = self.kernel_rows(xs, grad=(ncomp == 3), scale=sw, values=(ncomp == 1), hier=hier)
            rows = val if ncomp == 1 else dval
            st, en = self._site_ranges(ks)
            S = sets[nsets]
            S.n, S.ncomp, S.weight = xs.shape[0], ncomp, 1.0
            S.val = ptr(rows)
            tgt = None
            if target is not None:
                tgt = target.to(dev, torch.float32)
                tgt = ((tgt[perm] if perm is not None else tgt) * sw).contiguous()
                S.target = ptr(tgt)
            for d in range(self.svh.depth):
                S.start[d], S.end[d] = ptr(st[d]), ptr(en[d])
            keep += [xs, rows, st, en, tgt, ks]
            nsets += 1
        # structure pass: own-upper counts + in-degrees -> exclusive scans -> final CSR row pointers
        counts = torch.zeros((4, M + 1), dtype=torch.int32, device=dev)
        rowcount, crosscount, samelow, indeg = counts[0], counts[1], counts[2], counts[3]
        ws = torch.empty(int(_lib.lib.nksr_assemble_workspace_bytes(C.byref(hier))), dtype=torch.uint8, device=dev)
        td = _tick('_', time.perf_counter())
        call('nksr_assemble_count', C.byref(hier), ptr(ws), ptr(rowcount), ptr(crosscount), ptr(samelow), ptr(indeg), stream())
        n_up, n_mir = [int(v) for v in counts[:2].sum(dim=1, dtype=torch.int64).tolist()]
        td = _tick('asm:count', td)
        nnz = 2 * n_up + M
        if nnz >= 2 ** 31 - 4096:
            raise RuntimeError('assembled system too large for one chunk (M=%d, nnz=%d >= 2^31): use the matrix-free solve '
                               '(fused_mode=True) or pass chunk_size= to reconstruct() (examples/recons_by_chunk.py)' % (M, nnz))
        rowlen = indeg + samelow + rowcount + 1     # [cross-level mirrors][same-level lower][own upper][diagonal]
        rowlen[M] = 0
        rowptr = ops.exclusive_sum_i32(rowlen)
        mir_off = ops.exclusive_sum_i32(crosscount)
        mirptr = ops.exclusive_sum_i32(indeg)
        col_bits = ops._bits(M)
        # physical (tile-interleaved, zero-padded) CSR arrays for the streaming SpMV: packed 21-bit columns
        # (6.67 bytes per entry) whenever the unknowns fit, int32 columns otherwise (include/nksr_hip.h)
        fmt = 1 if M <= (1 << 21) and int(self.solver_config.get('col_format', 1)) == 1 else 0
        if coarse_from is not None:
            fmt = 2
        chunk, tile = (4608, 192) if fmt == 1 else ((4096, 256) if fmt == 0 else (1, 1))
        npad = (nnz + chunk - 1) // chunk * chunk
        cols = torch.empty(npad, dtype=torch.int32, device=dev)
        vals = torch.empty(npad, dtype=torch.float32, device=dev)
        # only the pad must be zero (valid column 0, value 0); the last tile is interleaved, so its
        # unwritten slots are scattered through the whole tile: clear it from its start
        tail = nnz // tile * tile
        cols[tail:].zero_()
        vals[tail:].zero_()
        diag = torch.empty(M, dtype=torch.float32, device=dev)
        b = torch.empty(M, dtype=torch.float32, device=dev)
        mir_k = torch.empty(n_mir, dtype=torch.int64, device=dev)
        mir_v = torch.empty(n_mir, dtype=torch.float32, device=dev)
        # coarse cells hold thousands of site rows: their Gram blocks are accumulated by several wavefronts each (csrc/assemble.hip)
        split_bytes = int(_lib.lib.nksr_assemble_split_bytes(C.byref(hier), sum(int(sets[i].n) * int(sets[i].ncomp) for i in range(nsets))))
        split = torch.empty(split_bytes, dtype=torch.uint8, device=dev) if split_bytes else None
        call('nksr_assemble', C.byref(hier), sets, nsets, float(reg_weight), col_bits, ptr(ws), ptr(rowptr), ptr(indeg),
             ptr(samelow), ptr(mir_off), fmt, ptr(cols), ptr(vals), ptr(diag), ptr(mir_k), ptr(mir_v), ptr(b), ptr(split), split_bytes, stream())
        td = _tick('asm:blocks+fill', td)
        del split
        del ws
        ks, vs = ops.sort_pairs(mir_k, mir_v.view(torch.int32), end_bit=col_bits)   # stable, destination-row bits only
        del mir_k, mir_v
        call('nksr_place_mirrors', ptr(ks), ptr(vs.view(torch.float32)), n_mir, col_bits, ptr(rowptr), ptr(mirptr), fmt, ptr(cols),
             ptr(vals), stream())
        del ks, vs
        td = _tick('asm:mirrors', td)
        if fmt == 1:
            packed = torch.empty(npad // 3, dtype=torch.int64, device=dev)
            call('nksr_pack_cols21', ptr(cols), npad, ptr(packed), stream())
            cols = packed
        if coarse_from is None:
            self.nnz = nnz
        del keep
        return rowptr, cols, vals, diag, b

    # ---- solve ------------------------------------------------------------------------------------------
    def solve_non_fused(self, pos_xyz, normal_xyz, normal_value, pos_weight, normal_weight, reg_weight=1.0,
                        pos_sorted_keys=None, normal_sorted_keys=None):
        """Assemble the sparse system explicitly and solve it with Jacobi-PCG."""
        from .. import solver
        t0 = time.perf_counter()
        rowptr, cols, vals, diag, b = self.assemble(pos_xyz, normal_xyz, normal_value, pos_weight, normal_weight, reg_weight,
                                                    pos_sorted_keys, normal_sorted_keys)
        # the same preconditioner policy as the matrix-free solve: hierarchies of 5+ levels (Jacobi needs ~47 iterations per
        # tree_depth-5 chunk, the coarse-level block ~13) or an explicit solver_config['coarse_precond']
        cfg = self.solver_config
        pc = None
        if cfg.get('coarse_precond') is not False and (isinstance(cfg.get('coarse_precond'), dict) or self.svh.depth >= 5):
            pc = self._coarse_precond(None, reg_weight, sites=dict(pos_xyz=pos_xyz, normal_xyz=normal_xyz, normal_value=normal_value,
                                                                   pos_weight=pos_weight, normal_weight=normal_weight,
                                                                   pos_sorted_keys=pos_sorted_keys, normal_sorted_keys=normal_sorted_keys))
        if cfg.get('verbose') or cfg.get('sync_timing'):
            torch.cuda.current_stream().synchronize()
        t1 = time.perf_counter()
        x, iters, rel = solver.pcg_solve(rowptr, cols, vals, diag, b, tol=cfg['tol'], max_iter=cfg['max_iter'], check_every=cfg['check_every'],
                                         precond=pc['pc'] if pc else None)
        t2 = time.perf_counter()
        self.alpha = x
        self.matrix = (rowptr, cols, vals, diag)
        self._fused_op = None
        self._pc = pc if self._wants_grad(normal_value) else None      # the adjoint solve of the backward pass takes the same preconditioner (_solve_system)
        self.rhs, self.diag = b, diag
        self.solve_info = {'iters': iters, 'rel_residual': rel, 'M': int(b.numel()), 'nnz': int(self.nnz),
                           'coarse_precond': ({k: pc[k] for k in ('first_level', 'unknowns', 'nnz', 'steps', 'lambda', 'gershgorin')} if pc else None),
                           'jacobi_fallbacks': solver.last_fallbacks, 't_assemble': t1 - t0, 't_pcg': t2 - t1}
        self._attach_autograd(pos_xyz, normal_xyz, normal_value, pos_weight, normal_weight)
        if self.solver_config.get('verbose'):
            print('[KernelField] M=%d nnz=%d iters=%d rel=%.3e assemble=%.3fs pcg=%.3fs' % (
                b.numel(), self.nnz, iters, rel, t1 - t0, t2 - t1))
        return self

    # ---- matrix-free ("fused") solve ---------------------------------------------------------------------
    def fused_operator(self, pos_xyz, normal_xyz, normal_value, pos_weight, normal_weight, pos_sorted_keys=None, normal_sorted_keys=None,
                       pos_value=None, segments=None):
        """Everything the matrix-free operator needs (csrc/fused.hip, nksr_fused_op_t): the level-major kernel rows of both
        site sets in one array (pre-multiplied by sqrt(weight)), their targets, the global neighbour table and the work
        items.  Returns a dict; ``keep`` holds the buffers the C struct points into."""
        dev = self.device
        svh = self.svh
        M, L = svh.num_unknowns, svh.depth
        if M == 0:
            raise RuntimeError('empty hierarchy')
        fac = self._row_format() == 'factors'
        specs = []
        # (ncomp = ROWS a site owns in the list: a normal site three -- or, in the factor form, four: a header row that carries phi
        # and contributes nothing, then one row per axis)
        for xyz, target, weight, ncomp, pre in ((pos_xyz, pos_value, pos_weight, 1, pos_sorted_keys),
                                                 (normal_xyz, normal_value, normal_weight, 4 if fac else 3, normal_sorted_keys)):
            if xyz is None or xyz.shape[0] == 0:
                continue
            per_site = torch.is_tensor(weight)            # batched chunks: every site carries its own chunk's sqrt(weight)
            if not per_site and not float(weight) >= 0.0:
                raise RuntimeError('solver weights must be >= 0')
            xyz = xyz.to(dev, torch.float32).contiguous()
            if pre is not None:
                ks, perm, xs = pre, None, xyz
            else:
                ks, perm = self._sorted_sites(xyz)
                xs = xyz[perm].contiguous()
            if per_site:
                sw = weight.to(dev, torch.float32)
                sw = (sw[perm] if perm is not None else sw).contiguous()
            else:
                sw = float(weight) ** 0.5
            specs.append((xs, ks, perm, target, sw, ncomp))
        if not specs:
            raise RuntimeError('no constraint sites')
        # ONE Morton-ordered row list for all site sets (stable sort of the sites' level-0 keys; a position site owns one row, a
        # normal site three): the rows of a cell -- of both sets -- are then one contiguous run at every level
        counts_s = [sp[0].shape[0] for sp in specs]
        rows_total = sum(n * sp[5] for n, sp in zip(counts_s, specs))
        nsite = sum(counts_s)
        pad_rows = item_seg = None
        if len(specs) == 1 and segments is None:
            row_index = [torch.arange(counts_s[0], dtype=torch.int32, device=dev) * specs[0][5]]
        elif len(specs) == 2 and os.environ.get('NKSR_ROW_ORDER', 'merge') != 'sort':
            # Both site lists are sorted already: the merged (stable: set 0 first on equal keys) order is a MERGE, and all it is needed
            # for is every site's first row = its own sites before it + the other set's sites before it -- two rank passes
            # (nksr_rank_sorted) instead of a 63-bit radix sort of the concatenated keys, a scan and a scatter
            (xa, ka, _, _, _, ca), (xb, kb, _, _, _, cb) = specs
            na, nb = counts_s
            ra = torch.empty(na, dtype=torch.int32, device=dev)
            rb_ = torch.empty(nb, dtype=torch.int32, device=dev)
            call('nksr_rank_sorted', ptr(kb), nb, ptr(ka), na, 0, ptr(ra), stream())          # sites of set 1 with a smaller key
            call('nksr_rank_sorted', ptr(ka), na, ptr(kb), nb, 1, ptr(rb_), stream())         # sites of set 0 with a smaller or equal key
            fa = torch.arange(na, dtype=torch.int32, device=dev) * ca + ra * cb
            fb = torch.arange(nb, dtype=torch.int32, device=dev) * cb + rb_ * ca
            if segments is not None:
                klo = segments.key_lo
                rb = (torch.searchsorted(ka, klo) * ca + torch.searchsorted(kb, klo) * cb).long()
                rb = torch.cat([rb, rb.new_tensor([rows_total])])                                               # unpadded row bounds
                rows_seg = rb[1:] - rb[:-1]
                pad = (-rows_seg) % 256
                pad_before = torch.cumsum(pad, 0) - pad
                pb32 = pad_before.to(torch.int32)
                fa = fa + pb32[segments.of_keys(ka)]
                fb = fb + pb32[segments.of_keys(kb)]
                ends = rb[1:] + pad_before
                pad_rows = (ends[:, None] + torch.arange(255, device=dev)[None])[torch.arange(255, device=dev)[None] < pad[:, None]]
                rows_total = rows_total + int(pad.sum().item())
                item_start = (rb[:-1] + pad_before) // 32
                item_seg = (torch.bucketize(torch.arange(rows_total // 32 + 2, device=dev), item_start, right=True) - 1).clamp_(0, segments.nseg - 1).to(torch.int32)
            row_index = [fa, fb]
        else:
            ks_all, order = ops.sort_pairs(torch.cat([sp[1] for sp in specs]), torch.arange(nsite, dtype=torch.int32, device=dev), level=0)
            order = order.long()
            ncomp_site = torch.cat([torch.full((n,), sp[5], dtype=torch.int32, device=dev) for n, sp in zip(counts_s, specs)])
            first_row = ops.exclusive_sum_i32(torch.cat([ncomp_site[order], ncomp_site.new_zeros(1)]))       # [nsite + 1]
            if segments is not None:
                # every segment's rows start on a workgroup boundary of the sweep (256 rows): which cells meet in a workgroup, the partial
                # blocks of the others -- and with them every summation order of the operator -- are then the same whether the
                # segment is solved alone or with others.  Pad rows have no cell (row_cells = -1) and zero values.
                sb = torch.searchsorted(ks_all, torch.cat([segments.key_lo, segments.key_hi[-1:]]))            # site bounds [nseg + 1]
                sb[-1] = nsite
                rb = first_row[sb].long()                                                                       # unpadded row bounds
                rows_seg = rb[1:] - rb[:-1]
                pad = (-rows_seg) % 256
                pad_before = torch.cumsum(pad, 0) - pad
                first_row = first_row[:nsite] + pad_before[segments.of_keys(ks_all)].to(torch.int32)
                ends = rb[1:] + pad_before                                                                      # first pad row of every segment
                pad_rows = (ends[:, None] + torch.arange(255, device=dev)[None])[torch.arange(255, device=dev)[None] < pad[:, None]]
                rows_total = rows_total + int(pad.sum().item())
                # segment of every 32-row work item (the solve skips the items of converged segments)
                item_start = (rb[:-1] + pad_before) // 32
                item_seg = (torch.bucketize(torch.arange(rows_total // 32 + 2, device=dev), item_start, right=True) - 1).clamp_(0, segments.nseg - 1).to(torch.int32)
            else:
                first_row = first_row[:nsite]
            row_of_site = torch.empty(nsite, dtype=torch.int32, device=dev)
            row_of_site[order] = first_row
            row_index = list(torch.split(row_of_site, counts_s))
        td = _tick('_', time.perf_counter())
        pad = 320 * 27     # (the operator's loads are unconditional: the last workgroup reads up to 255 + 63 rows past the end)
        # kernel_dim 4, dense-slot rows: ONE launch writes the rows of both sets (csrc/rows.hip: k_kernel_rows_merged -- the interleaved
        # rows of two launches reach HBM as partial lines).  NKSR_ROWS_KERNEL=site keeps the launch per set (bit-identical rows).
        # NKSR_ROWS_LAYOUT=compact (opt-in, round 6): only the slots of a cell's EXISTING neighbours are stored and streamed.  Built for
        # the "quarter of structural zeros" the round-5 review quoted for the 64-chunk scene -- measured there: 3.3 % fewer words (absent
        # neighbours are 4 % of the slots of a 26-connected band; the rest of the physical-vs-algorithmic gap is tables, not zeros), and
        # more than 2^31 16-byte units in one batch.  It pays on thin structures only; the default stays the 27-slot row.
        merged = (not fac and self.kdim == 4 and self.hidden in (16, 32) and os.environ.get('NKSR_ROWS_KERNEL', 'merged') == 'merged'
                  and max(counts_s) < 2 ** 29)
        compact = merged and os.environ.get('NKSR_ROWS_LAYOUT', 'dense') == 'compact'
        rows_all = fac_vec = fac_pos = psi_all = None
        row_cells = torch.empty((L, rows_total), dtype=torch.int32, device=dev)
        targets_all = torch.zeros(rows_total + 320, dtype=torch.float32, device=dev)[:rows_total]      # (readable past the end, like the rows)
        keep = [targets_all, row_cells]
        span = torch.empty((3, M), dtype=torch.int32, device=dev)          # first / last row of every cell, first workgroup
        counts = torch.empty(M + 1, dtype=torch.int32, device=dev)
        item_begin = torch.empty(int(_lib.lib.nksr_fused_item_entries(rows_total)), dtype=torch.int32, device=dev)
        nbr32 = torch.empty((M, 32), dtype=torch.int32, device=dev)
        nbrT = torch.empty((27, M), dtype=torch.int32, device=dev)

        def tables(rowbase4=None):
            # work items = runs of 32 rows, eight of them a workgroup of the sweep; a cell whose rows lie inside one workgroup is finished
            # there, a cell that reaches into k > 1 workgroups owns k partial blocks (the coarse cells: ~1 % of all)
            call('nksr_fused_block_counts', L, M, rows_total, ptr(row_cells), ptr(span), ptr(item_begin), ptr(counts), stream())
            return ops.exclusive_sum_i32(counts)

        rows_words = 0
        if merged:
            row_src = torch.full((rows_total,), -1, dtype=torch.int32, device=dev)
            args = {1: (None, None, 1.0), 3: (None, None, 1.0)}
            for (xs, ks, perm, target, sw, ncomp), ri in zip(specs, row_index):
                ri = ri.contiguous()
                call('nksr_row_sources', ptr(ri), xs.shape[0], ncomp, 0 if ncomp == 1 else 1, ptr(row_src), stream())
                tw = torch.is_tensor(sw)
                args[ncomp] = (xs, sw if tw else None, 1.0 if tw else sw)
                keep += [xs, ri]
            (xa, sa, fa_), (xb, sb, fb_) = args[1], args[3]
            keep.append(row_src)
            if compact:
                # cells of all rows first: the cells' row spans decide where their rows lie in the compact array
                call('nksr_row_cells_merged', C.byref(self._hier), ptr(xa), ptr(xb), ptr(row_src), rows_total, ptr(row_cells), stream())
                offsets = tables()
                sizes4 = torch.empty(M + 1, dtype=torch.int64, device=dev)
                call('nksr_fused_row_sizes', C.byref(self._hier), ptr(span), ptr(sizes4), stream())
                ends4 = torch.cumsum(sizes4, 0)
                nblocks, total4 = [int(v) for v in torch.stack([offsets[M].long(), ends4[M]]).tolist()]          # (ONE readback)
                if total4 + 1 >= 2 ** 31:
                    raise RuntimeError('too many kernel-row entries for one solve (%d x 16 bytes): pass chunk_size= / a smaller chunk_batch_points' % total4)
                rowbase4 = (ends4[:M] - sizes4[:M] + 1).to(torch.int32)                                          # (unit 0 = the zero block)
                rows_words = 4 * (total4 + 1)
                rows_all = torch.empty(rows_words + pad, dtype=torch.float32, device=dev)
                rows_all[:4].zero_()
                rows_all[rows_words:].zero_()
                call('nksr_fused_tables', C.byref(self._hier), rows_total, ptr(item_begin), ptr(offsets), ptr(span), ptr(rowbase4), ptr(nbr32), ptr(nbrT), stream())
                call('nksr_kernel_rows_merged', C.byref(self._hier), ptr(xa), ptr(sa), float(fa_), ptr(xb), ptr(sb), float(fb_),
                     int(self.approx_kernel_grad), ptr(row_src), rows_total, ptr(row_cells), 1, ptr(nbr32), ptr(rows_all), stream())
                del sizes4, ends4, rowbase4
        if fac:
            fac_vec = torch.empty(L * rows_total * 4 + 320 * 4, dtype=torch.float32, device=dev)
            fac_vec[L * rows_total * 4:].zero_()
            fac_pos = torch.empty((rows_total + 320) * 4, dtype=torch.float32, device=dev)
            fac_pos[rows_total * 4:].zero_()
            psi_all = torch.cat([p.reshape(-1, 4) for p in self._psi]).contiguous()
            assert psi_all.shape[0] == M
        elif not compact:
            rows_all = torch.empty(L * rows_total * 27 + pad, dtype=torch.float32, device=dev)
            rows_all[L * rows_total * 27:].zero_()
        if not compact and pad_rows is not None and pad_rows.numel():
            row_cells[:, pad_rows] = -1
            if fac:
                fac_vec[:L * rows_total * 4].view(L, rows_total, 4)[:, pad_rows] = 0.0
                fac_pos[:rows_total * 4].view(rows_total, 4)[pad_rows] = 0.0                             # (kind 0: a position row without a cell)
            else:
                rows_all[:L * rows_total * 27].view(L, rows_total, 27)[:, pad_rows] = 0.0
        keep += [rows_all, fac_vec, fac_pos, psi_all]
        td = _tick('op:alloc', td)
        if merged and not compact:
            # the rows' cells first (one pass, the probes of all levels in flight together): the row kernel then starts from them
            pre = os.environ.get('NKSR_ROWS_PRECELLS', '1') != '0'
            if pre:
                call('nksr_row_cells_merged', C.byref(self._hier), ptr(xa), ptr(xb), ptr(row_src), rows_total, ptr(row_cells), stream())
            call('nksr_kernel_rows_merged', C.byref(self._hier), ptr(xa), ptr(sa), float(fa_), ptr(xb), ptr(sb), float(fb_),
                 int(self.approx_kernel_grad), ptr(row_src), rows_total, ptr(row_cells), int(pre), None, ptr(rows_all), stream())
        for (xs, ks, perm, target, sw, ncomp), ri in zip(specs, row_index):
            ri = ri.contiguous()
            tensor_w = torch.is_tensor(sw)
            if merged:
                pass
            elif fac:
                self.kernel_factors_level_major(xs, ncomp == 4, 1.0 if tensor_w else sw, fac_vec, fac_pos, rows_total, ri, row_cells,
                                                site_scale=sw if tensor_w else None)
            else:
                self.kernel_rows_level_major(xs, ncomp == 3, 1.0 if tensor_w else sw, rows_all, rows_total, ri, row_cells,
                                             site_scale=sw if tensor_w else None)
            if target is not None:
                nc = 3 if ncomp >= 3 else 1                                                             # target components; the header row's is 0
                tgt = target.detach().to(dev, torch.float32)
                tgt = (tgt[perm] if perm is not None else tgt).reshape(xs.shape[0], nc)
                tgt = tgt * (sw[:, None] if tensor_w else sw)                                           # row order (site, component)
                targets_all[(ri.long()[:, None] + (ncomp - nc) + torch.arange(nc, device=dev)[None]).reshape(-1)] = tgt.reshape(-1)
            keep += [xs, ri]
        td = _tick('op:kernel_rows', td)
        if not compact:
            offsets = tables()
            call('nksr_fused_tables', C.byref(self._hier), rows_total, ptr(item_begin), ptr(offsets), ptr(span), None, ptr(nbr32), ptr(nbrT), stream())
            nblocks = int(offsets[M].item())
        big = torch.nonzero(counts[:M] > 16).reshape(-1).to(torch.int32)          # coarse cells: a workgroup each in the per-cell sum
        multi = torch.cat([big, torch.nonzero((counts[:M] > 1) & (counts[:M] <= 16)).reshape(-1).to(torch.int32)])
        ws = torch.empty(int(_lib.lib.nksr_fused_workspace_bytes(nblocks, M)), dtype=torch.uint8, device=dev)
        cell_sums = torch.zeros((27, M), dtype=torch.float32, device=dev)
        op = FusedOpT()
        op.depth, op.M, op.n_multi, op.n_big, op.rows_total, op.nblocks = L, M, int(multi.numel()), int(big.numel()), rows_total, nblocks
        op.rows_all, op.targets_all, op.row_cells, op.nbr32, op.nbrT = ptr(rows_all), ptr(targets_all), ptr(row_cells), ptr(nbr32), ptr(nbrT)
        op.compact, op.rows_words = int(compact), int(rows_words)
        if fac:
            op.fac_vec, op.fac_pos, op.psi_all, op.inv_w0 = ptr(fac_vec), ptr(fac_pos), ptr(psi_all), float(svh.inv_w0)
        op.item_begin = ptr(item_begin)
        op.offsets, op.multi, op.workspace, op.cell_sums = ptr(offsets), (ptr(multi) if multi.numel() else None), ptr(ws), ptr(cell_sums)
        # SURVEY.md section 8d counts the operator's bytes per STORED entry; the dense-slot rows hold structural zeros (absent
        # neighbours, B-spline support ends): the set-up pass counts the non-zero slots on its way (read back on demand)
        nnz_counter = torch.zeros(1, dtype=torch.int64, device=dev)
        op.nnz_counter = ptr(nnz_counter)
        if item_seg is not None:
            op.item_seg, op.unknown_seg = ptr(item_seg), ptr(segments.unknown_seg)
            keep += [item_seg, segments.unknown_seg]
        keep += [nbr32, nbrT, item_begin, offsets, multi, ws, cell_sums, nnz_counter]
        td = _tick('op:tables', td)
        return {'op': op, 'nsets': len(specs), 'nblocks': nblocks, 'rows_total': rows_total, 'n_multi': int(multi.numel()),
                'nnz_counter': nnz_counter, 'keep': keep, 'span': span, 'rows_all': rows_all, 'row_format': 'factors' if fac else 'dense',
                'compact': bool(compact), 'rows_words': rows_words, 'nbr32': nbr32,
                'fac_vec': fac_vec, 'fac_pos': fac_pos, 'row_cells': row_cells, 'targets_all': targets_all}

    def dense_rows(self, op):
        """[L, rows_total, 27] dense-slot rows of a matrix-free operator, whatever its layout (test / export helper)."""
        L, R = self.svh.depth, op['rows_total']
        if not op.get('compact'):
            return op['rows_all'][:L * R * 27].view(L, R, 27)
        out = torch.zeros((L, R, 27), dtype=torch.float32, device=self.device)
        cells = op['row_cells'].long()
        tb = op['nbr32'].long()
        rr = torch.arange(R, device=self.device)
        for d in range(L):
            c = cells[d]
            ok = c >= 0
            cj = c.clamp(min=0)
            mask, first, b4 = tb[cj, 31], tb[cj, 28], tb[cj, 30]
            k = torch.zeros_like(mask)
            for q in range(27):
                k += (mask >> q) & 1
            rank = torch.zeros_like(mask)
            for q in range(27):
                pres = ok & (((mask >> q) & 1) == 1)
                idx = b4 * 4 + (rr - first) * k + rank
                out[d, pres, q] = op['rows_all'][idx[pres]]
                rank = rank + ((mask >> q) & 1)
        return out

    def fused_rhs_diag(self, op, reg_weight=1.0, dense_from=None):
        """Right-hand side and Jacobi diagonal from ONE set-up sweep.  ``dense_from`` = c0 (factor form only): the sweep also leaves
        the rebuilt rows of the levels >= c0 in op['dense'] -- what the coarse-level block of the preconditioner is assembled from."""
        M = self.svh.num_unknowns
        b = torch.empty(M, dtype=torch.float32, device=self.device)
        diag = torch.empty(M, dtype=torch.float32, device=self.device)
        if dense_from is not None and op.get('row_format') == 'factors':
            self._arm_dense(op, int(dense_from))
        call('nksr_fused_rhs_diag', C.byref(op['op']), float(reg_weight), ptr(b), ptr(diag), stream())
        if op.get('dense') is not None:
            op['op'].dense_out = None          # (written; later sweeps must not write it again)
        return b, diag

    def _arm_dense(self, op, c0):
        L = self.svh.depth
        dense = torch.empty((L - c0, op['rows_total'], 27), dtype=torch.float32, device=self.device)
        op['dense'], op['dense_from'] = dense, c0
        op['op'].dense_from, op['op'].dense_out = c0, ptr(dense)
        return dense

    def _dense_coarse_rows(self, op, c0):
        """[L - c0, rows_total, 27] dense rows of the levels >= c0 of a factor-form operator (taken over: the operator forgets them)."""
        dense = op.get('dense')
        if dense is None or op.get('dense_from') != c0:
            dense = self._arm_dense(op, c0)
            call('nksr_fused_expand_rows', C.byref(op['op']), stream())
            op['op'].dense_out = None
        op['dense'] = None
        return dense

    def _small_field(self):
        """Single fields of at most 2^16 unknowns (configs[1], one scan of examples/recons_waymo_cpu.py): the policy takes the block of
        the levels >= 1 at once -- see solve_fused."""
        return 2 <= self.svh.depth < 5 and self.svh.num_unknowns <= SMALL_FIELD_UNKNOWNS      # (5+ levels: the block of the levels >= 2, as ever)

    def _pc_first_level(self, segments=None):
        """First level of the coarse-level block the solve is going to build at once, or None (see solve_fused / _coarse_precond)."""
        cfg = self.solver_config.get('coarse_precond')
        if cfg is False:
            return None
        auto = cfg is None and segments is None
        if auto and self.svh.depth < 5 and not self._small_field():
            return None
        cfg = cfg if isinstance(cfg, dict) else (dict(SMALL_FIELD_PC) if auto and self._small_field() else {})
        c0 = int(cfg.get('first_level', float(os.environ.get('NKSR_PC_LEVEL', 2))))
        off, M = self.svh.offsets, self.svh.num_unknowns
        return c0 if (0 < c0 < self.svh.depth and M - off[c0] >= 1) else None

    def stored_entries(self):
        """Non-zero entries of G and Q of the last matrix-free solve (counted by its diagonal pass; one small device read)."""
        t = getattr(self, '_stored_entries', None)
        return int(t.item()) if t is not None else None

    def fused_apply(self, op, x, reg_weight=1.0):
        """y = (w_p G^T G + w_n Q^T Q + reg I) x without the matrix (test / export helper)."""
        y = torch.empty_like(x)
        call('nksr_fused_apply', C.byref(op['op']), float(reg_weight), ptr(x.contiguous()), ptr(y), stream())
        return y

    def _coarse_precond(self, op, reg_weight, segments=None, sites=None, override=None):
        """Block preconditioner of the coarse levels (nksr_coarse_precond_t, csrc/pcg.hip): the diagonal block of the levels >= c0
        assembled as a small plain CSR + the largest Jacobi-scaled eigenvalue of every segment's block (left on the device: no
        host sync).  solver_config['coarse_precond']: None = automatic (see solve_fused), False = off, or a dict
        {'first_level', 'steps', 'ratio'}."""
        cfg = self.solver_config.get('coarse_precond')
        L = self.svh.depth
        if cfg is False:
            return None
        cfg = dict(cfg) if isinstance(cfg, dict) else dict(override or {})
        for k, e in (('first_level', 'NKSR_PC_LEVEL'), ('steps', 'NKSR_PC_STEPS'), ('ratio', 'NKSR_PC_RATIO')):      # tuning knobs
            if e in os.environ and k not in cfg:
                cfg[k] = float(os.environ[e])
        c0 = int(cfg.get('first_level', 2))
        off = self.svh.offsets
        M = self.svh.num_unknowns
        if not 0 < c0 < L or M - off[c0] < 1:
            return None
        n = M - off[c0]
        nseg = segments.nseg if segments is not None else 1
        td = _tick('_', time.perf_counter())
        if op is not None:      # from the kernel rows the matrix-free operator already holds
            rowptr, cols, vals, diag, _ = self.assemble(None, None, None, 1.0, 1.0, reg_weight, coarse_from=c0, fused_op=op)
        else:                   # the assembled solve: the same block from the site sets (rows of the masked hierarchy)
            rowptr, cols, vals, diag, _ = self.assemble(reg_weight=reg_weight, coarse_from=c0, **sites)
        td = _tick('pc:assemble', td)
        lam = torch.empty(nseg, dtype=torch.float32, device=self.device)
        coef = torch.empty(nseg * (1 + 2 * PC_MAX_STEPS), dtype=torch.float32, device=self.device)
        row_seg = segments.unknown_seg[off[c0]:].contiguous() if segments is not None else None
        pc = CoarsePrecondT()
        # the interval's upper end: 1.1 x the power-iteration estimate (a LOWER bound of lambda_max, within ~1 % after 8 steps), capped by
        # the Gershgorin bound (a true upper bound, 2-3x too large to be used by itself).  'lambda_scale' is a test knob: < 1 forces the
        # polynomial to lose definiteness, which the PCG answers by restarting the segment with Jacobi alone (csrc/pcg.hip)
        pc.first, pc.n, pc.steps, pc.lambda_scale, pc.ratio = off[c0], n, int(cfg.get('steps', 8)), float(cfg.get('lambda_scale', 1.1)), float(cfg.get('ratio', PC_RATIO))
        gersh = torch.empty(nseg, dtype=torch.float32, device=self.device)
        pc.lambda_, pc.coef = ptr(lam), ptr(coef)
        nnz = int(cols.numel())
        info = {'first_level': c0, 'unknowns': n, 'nnz': nnz, 'steps': int(pc.steps), 'lambda': lam}
        # packed form (csrc/pcg.hip, format 1): Jacobi-scaled half-precision values + 16-bit segment-local columns, 4 bytes per entry
        # instead of 8 -- when every segment holds fewer than 2^16 coarse unknowns (chunks do; a large single field does not)
        rs = row_seg if row_seg is not None else torch.zeros(n, dtype=torch.int32, device=self.device)
        counts = torch.bincount(rs.long(), minlength=nseg)
        if cfg.get('packed', os.environ.get('NKSR_PC_PACKED', '1') != '0') and int(counts.max()) < 65536:
            ar = torch.arange(n, dtype=torch.int64, device=self.device)
            old_of_new = torch.argsort(rs.long() * n + ar)
            new_of_old = torch.empty_like(old_of_new)
            new_of_old[old_of_new] = ar
            seg_base = torch.cat([counts.new_zeros(1), torch.cumsum(counts, 0)]).to(torch.int32)
            row_seg_new = rs[old_of_new].contiguous()
            o2n, n2o = old_of_new.to(torch.int32), new_of_old.to(torch.int32)
            drop = float(cfg.get('drop', os.environ.get('NKSR_PC_DROP', PC_DROP_TOL)))
            lens = torch.empty(n + 1, dtype=torch.int32, device=self.device)
            lens[n:] = 0
            call('nksr_coarse_pack_count', ptr(rowptr), ptr(cols), ptr(vals), ptr(diag), n, ptr(o2n), drop, ptr(lens), stream())
            prow = ops.exclusive_sum_i32(lens)
            kept = int(prow[n].item())
            info.update(nnz_kept=kept + n, drop=drop)
            packed = torch.empty(max(kept, 1), dtype=torch.int32, device=self.device)
            dis = torch.empty(n, dtype=torch.float32, device=self.device)
            call('nksr_coarse_pack', ptr(rowptr), ptr(cols), ptr(vals), ptr(diag), n, ptr(o2n), ptr(n2o), ptr(row_seg_new), ptr(seg_base), ptr(prow),
                 drop, ptr(packed), ptr(dis), stream())
            work = torch.empty(4 * n, dtype=torch.float32, device=self.device)
            # ten steps instead of eight on large blocks: a packed step costs a third of a plain one (four rows per wavefront, half the
            # bytes, no tails), and every PCG iteration saved is a sweep over all kernel rows (configs[4], one GPU, ratio 40:
            # 8 / 10 / 12 steps -> 11.19 / 10.91 / 10.72 iterations per chunk, 390.7 / 392.5 / 395.7 ms per step: flat);
            # small blocks are bound by the number of launches, not by bytes: they keep eight
            # (chunk mode always takes ten: the count must not depend on how many chunks share the batch -- a chunk's iterates are
            # the same bits alone and among 63 others, tests/test_gpu_full_size.py)
            if 'steps' not in cfg and (segments is not None or n >= 100000):
                pc.steps = 10
                info['steps'] = 10
            pc.format, pc.row_seg, pc.work = 1, ptr(row_seg_new), ptr(work)
            pc.packed, pc.packed_rowptr, pc.dis, pc.old_of_new, pc.seg_base = ptr(packed), ptr(prow), ptr(dis), ptr(o2n), ptr(seg_base)
            call('nksr_coarse_lambda_max_packed', C.byref(pc), nseg, 8, ptr(work), ptr(lam), stream())
            call('nksr_coarse_gershgorin', C.byref(pc), nseg, ptr(work), ptr(gersh), stream())
            pc.gersh = ptr(gersh)
            info.update(packed=True, gershgorin=gersh, keep=(packed, prow, dis, o2n, seg_base, row_seg_new, work, lam, coef, gersh))
            td = _tick('pc:pack+lambda', td)
            return dict(info, pc=pc)
        work = torch.empty(3 * n, dtype=torch.float32, device=self.device)
        # eight power-iteration steps from the all-ones vector land within ~1 % (measured): 10 % margin.  A segment whose block is
        # degenerate (no constraint rows on these levels: lambda <= 0) keeps Jacobi -- decided on the device (k_cheb_coeffs)
        call('nksr_coarse_lambda_max', ptr(rowptr), ptr(cols), ptr(vals), ptr(diag), n, 8, ptr(work), ptr(lam),
             C.byref(segments.c) if segments is not None else None, off[c0], stream())
        pc.format, pc.row_seg, pc.work = 0, ptr(row_seg), ptr(work)
        pc.rowptr, pc.cols, pc.vals, pc.diag = ptr(rowptr), ptr(cols), ptr(vals), ptr(diag)
        call('nksr_coarse_gershgorin', C.byref(pc), nseg, ptr(work), ptr(gersh), stream())
        pc.gersh = ptr(gersh)
        info.update(packed=False, gershgorin=gersh, keep=(rowptr, cols, vals, diag, work, lam, coef, row_seg, gersh))
        return dict(info, pc=pc)

    def solve_fused(self, pos_xyz, normal_xyz, normal_value, pos_weight, normal_weight, reg_weight=1.0,
                    pos_sorted_keys=None, normal_sorted_keys=None, segments=None):
        """Matrix-free Jacobi-PCG on the normal equations: no assembly, ~8 bytes per dense kernel-row slot per
        iteration (examples/recons_waymo.py:33 ``fused_mode=True``).  Same system, same stopping rule as
        solve_non_fused; the iterates agree to fp32 rounding (different summation order)."""
        t0 = time.perf_counter()
        td = _tick('_', t0)
        op = self.fused_operator(pos_xyz, normal_xyz, normal_value, pos_weight, normal_weight, pos_sorted_keys, normal_sorted_keys,
                                 segments=segments)
        td = _tick('fused_operator', td)
        dev = self.device
        M = self.svh.num_unknowns
        b, diag = self.fused_rhs_diag(op, reg_weight, dense_from=self._pc_first_level(segments))
        td = _tick('rhs_diag', td)
        cfg, tol = self.solver_config, float(self.solver_config['tol'])
        max_iter, check_every = int(cfg['max_iter']), int(cfg['check_every'])
        # Preconditioner policy (coarse_precond = None): hierarchies of 5+ levels get the coarse-level block at once (Jacobi alone
        # needs ~47 iterations per tree_depth-5 chunk); shallower ones start with Jacobi -- the 1M-point headline converges in 11
        # iterations, a set-up would not pay -- and switch after one unconverged round of check_every iterations (sparse /
        # sensor-only inputs and adaptive_depth 2 take 100+ Jacobi iterations at depth 4 too).
        # Batched chunk solves (``segments``) always take the block at once: a restart decided on the joint residual would make a
        # chunk's iterates depend on its batch mates.
        # Small single fields (<= 2^16 unknowns; round 6): the block of the levels >= 1 at once, eight steps on [lmax / 40, lmax] -- the
        # block is a few thousand unknowns there and costs little, and one failed Jacobi round was most of the solve (configs[1] 39 ->
        # 18 iterations, the 10 000-point bunny scan 93 -> 53, the smoke sphere 94 -> 57: tools/small_pc_sweep.py).
        auto = cfg.get('coarse_precond') is None and segments is None
        small = auto and self._small_field()
        if small:
            # an iteration of such a field is 16 launches of a few microseconds: the 14 no-op iterations behind convergence at 18 of
            # a 16-iteration round cost as much as 5 real ones -- the host looks every 6 (same iterates: convergence is per iteration on the device)
            check_every = min(check_every, SMALL_FIELD_CHECK_EVERY)
        pc = self._coarse_precond(op, reg_weight, segments, override=SMALL_FIELD_PC if small else None) if (not auto or small or self.svh.depth >= 5) else None
        op['dense'] = None                      # (dense coarse rows nobody took over)
        td = _tick('coarse_precond', td)
        if cfg.get('verbose') or cfg.get('sync_timing'):
            torch.cuda.current_stream().synchronize()
        t1 = time.perf_counter()
        nseg = segments.nseg if segments is not None else 1
        pws = torch.empty(int(_lib.lib.nksr_pcg_vector_workspace_bytes_seg(M, nseg, self.svh.depth if segments is not None else 1)),
                          dtype=torch.uint8, device=dev)

        def pcg(rhs, rtol, iters, precond):
            sol = torch.empty(M, dtype=torch.float32, device=dev)
            inf = (C.c_double * 3)()
            call('nksr_pcg_solve_fused', C.byref(op['op']), float(reg_weight), ptr(diag), ptr(rhs), ptr(sol), float(rtol), int(iters),
                 check_every, ptr(pws), C.byref(precond['pc']) if precond else None, C.byref(segments.c) if segments is not None else None,
                 inf, stream())
            # a segment whose Chebyshev block lost definiteness (r.z <= 0: eigenvalue bound too small) restarts with Jacobi alone on
            # the device (csrc/pcg.hip: k_spcg_pupdate) -- counted, not raised; only r.z <= 0 with Jacobi itself / NaN is an error
            fallbacks[0] += int(inf[2])
            if inf[1] < 0:
                raise RuntimeError('PCG breakdown (r.z <= 0 with the Jacobi preconditioner after %d iterations, relative residual %.3e): '
                                   'the system is not positive definite (non-finite rows or weights?)' % (int(inf[0]), -inf[1]))
            return sol, int(inf[0]), float(inf[1])
        fallbacks = [0]
        if pc is not None or not auto or max_iter <= check_every:
            x, iters, rel = pcg(b, tol, max_iter, pc)
        else:
            x, iters, rel = pcg(b, tol, check_every, None)
            if rel > tol:
                # restart on the residual with the block preconditioner: A e = b - A x to the remaining accuracy
                pc = self._coarse_precond(op, reg_weight)
                r = b - self.fused_apply(op, x, reg_weight)
                bn, rn = float(torch.linalg.vector_norm(b.double())), float(torch.linalg.vector_norm(r.double()))
                if rn > tol * bn:
                    e, it2, rel2 = pcg(r, tol * bn / rn, max_iter - iters, pc)
                    x, iters, rel = x + e, iters + it2, rel2 * rn / bn
                else:
                    rel = rn / bn
        info = (float(iters), rel)
        t2 = time.perf_counter()
        if fallbacks[0]:
            import warnings
            warnings.warn('%d segment(s) of the solve restarted with the Jacobi preconditioner alone (the coarse-level block lost '
                          'definiteness: eigenvalue bound too small); the result is valid, the iteration count is higher' % fallbacks[0])
        self.alpha = x
        self.matrix = None
        self._fused_op, self._fused_reg = op, float(reg_weight)
        self._pc = pc if segments is None else None      # (a batched solve's block needs its segments: not kept)
        self._stored_entries = op['nnz_counter']
        self.rhs, self.diag = b, diag
        self.nnz = 0
        self.solve_info = {'iters': int(info[0]), 'rel_residual': float(info[1]), 'M': int(M), 'nnz': 0, 'fused': True,
                           'kernel_row_slots': 27 * self.svh.depth * op['rows_total'], 'partial_blocks': op['nblocks'],
                           'coarse_precond': ({k: pc[k] for k in ('first_level', 'unknowns', 'nnz', 'steps', 'lambda', 'gershgorin')} if pc else None),
                           'segments': nseg, 'segment_info': segments.info if segments is not None else None,
                           'jacobi_fallbacks': fallbacks[0], 't_assemble': t1 - t0, 't_pcg': t2 - t1}
        if self.solver_config.get('verbose'):
            print('[KernelField] fused: M=%d rows=%d iters=%d rel=%.3e rows+rhs=%.3fs pcg=%.3fs' % (
                M, op['rows_total'], int(info[0]), float(info[1]), t1 - t0, t2 - t1))
        if not self._wants_grad(normal_value):
            self._fused_op = self._pc = None          # only the backward pass needs the operator again: do not pin ~2 GB of rows
        self._attach_autograd(pos_xyz, normal_xyz, normal_value, pos_weight, normal_weight)
        return self

    # ---- differentiable solve (training path, models/nksr_net.py:105-112) ---------------------------------------
    def _solve_system(self, rhs):
        """A^-1 rhs with the system of the last solve (assembled CSR or matrix-free operator), same tolerance, same
        preconditioner (the coarse-level block of the forward solve, when it had one)."""
        from .. import solver
        cfg = self.solver_config
        pc = getattr(self, '_pc', None)
        if self.matrix is not None:
            rowptr, cols, vals, diag = self.matrix
            return solver.pcg_solve(rowptr, cols, vals, diag, rhs.contiguous(), tol=cfg['tol'], max_iter=cfg['max_iter'], check_every=cfg['check_every'],
                                    precond=pc['pc'] if pc else None)[0]
        if getattr(self, '_fused_op', None) is None:
            raise RuntimeError('the system of the last solve is gone: call solve*() under torch.enable_grad() with normal_value.requires_grad')
        M = self.svh.num_unknowns
        x = torch.empty(M, dtype=torch.float32, device=self.device)
        pws = torch.empty(int(_lib.lib.nksr_pcg_vector_workspace_bytes(M)), dtype=torch.uint8, device=self.device)
        info = (C.c_double * 3)()
        call('nksr_pcg_solve_fused', C.byref(self._fused_op['op']), self._fused_reg, ptr(self.diag), ptr(rhs.contiguous()), ptr(x), float(cfg['tol']),
             int(cfg['max_iter']), int(cfg['check_every']), ptr(pws), C.byref(pc['pc']) if pc else None, None, info, stream())
        if info[1] < 0:
            raise RuntimeError('PCG breakdown in the adjoint solve (r.z <= 0 with the Jacobi preconditioner): the system is not positive definite')
        return x

    def _theta(self):
        """The caller's basis-feature tensors and interpolator parameters that take part in an autograd graph.  The switch is the
        FEATURES: a field whose basis features carry no graph (inference: they come out of the HIP U-Net; tests: constants) stays
        out of autograd in theta even though an nn.Module's parameters require grad by default."""
        th = [f for f in self._feat_in if torch.is_tensor(f) and f.requires_grad]
        if not th:
            return []
        for m in self._interps_in:
            if isinstance(m, torch.nn.Module):
                th += [q for q in m.parameters() if q.requires_grad]
        return th

    def _wants_grad(self, normal_value):
        return torch.is_grad_enabled() and ((torch.is_tensor(normal_value) and normal_value.requires_grad) or len(self._theta()) > 0)

    def _attach_autograd(self, pos_xyz, normal_xyz, normal_value, pos_weight, normal_weight):
        """Under autograd, alpha becomes a differentiable function of the normal targets, the basis features and the interpolator
        weights by implicit differentiation: one more PCG solve with the same system in backward, then (for the features /
        weights) the vector-Jacobian product through the torch statement of the kernel rows (fields/kernel_rows_torch.py)."""
        if not self._wants_grad(normal_value):
            return
        nv = normal_value if torch.is_tensor(normal_value) else torch.zeros((0, 3), device=self.device)
        self.alpha = _SolveFunction.apply(self, self.alpha, None if pos_xyz is None else pos_xyz.detach(),
                                          None if normal_xyz is None else normal_xyz.detach(), nv, float(pos_weight), float(normal_weight),
                                          *self._theta())

    def _theta_vjp(self, sets, alpha, lam=None):
        """sum_r g_r . dR'_r / dtheta for the site sets ``sets`` = [(xyz, gradient rows?, sqrt weight, per-row coefficient fn)]:
        the coefficient function maps (u = R' alpha [, v = R' lambda]) of a set to the row factors (a, b) of
        g_r = a_r lambda + b_r alpha  (solve)  or  g_r = a_r alpha  (evaluation)."""
        from . import kernel_rows_torch as krt
        theta = self._theta()
        if not theta:
            return []
        if os.environ.get('NKSR_THETA_VJP', 'hip') != 'torch':
            # the product path: HIP kernels.  (The torch statement below is the REFERENCE the tests differentiate -- reached only
            # with NKSR_THETA_VJP=torch; it is never a fallback.)
            if any(torch.is_tensor(sw) for _, _, sw, _ in sets):
                raise RuntimeError('the backward pass of a batched chunk solve (per-site weights) is not supported: train on single fields')
            return self._theta_vjp_hip(sets, alpha, lam)
        with torch.enable_grad():
            S = torch.zeros((), dtype=torch.float32, device=self.device)
            for xyz, grad_rows, sw, coeff in sets:
                if xyz is None or xyz.shape[0] == 0:
                    continue
                R, idx = krt.rows(self.svh, self._interps_in, [f if torch.is_tensor(f) else torch.zeros((0, self.kdim), device=self.device)
                                                              for f in self._feat_in], xyz.to(self.device, torch.float32), grad_rows,
                                  self.approx_kernel_grad, scale=sw)
                with torch.no_grad():
                    Rd = R.detach()
                    u = krt.apply_rows(Rd, idx, alpha, grad_rows)
                    v = krt.apply_rows(Rd, idx, lam, grad_rows) if lam is not None else None
                    a, b = coeff(u, v)
                    m = (idx >= 0).to(torch.float32)
                    ag = alpha[idx.clamp(min=0)] * m                                   # [n, L, 27]
                    lg = lam[idx.clamp(min=0)] * m if lam is not None else None
                    if grad_rows:                                                      # rows [n, 3, L, 27], factors [n, 3]
                        g = (a[..., None, None] * lg[:, None] if lg is not None else 0.0) + b[..., None, None] * ag[:, None]
                    else:
                        g = (a[:, None, None] * lg if lg is not None else 0.0) + b[:, None, None] * ag
                S = S + (R * g).sum()
            grads = torch.autograd.grad(S, theta, allow_unused=True)
        return [gr if gr is not None else torch.zeros_like(t) for gr, t in zip(grads, theta)]

    def _theta_vjp_hip(self, sets, alpha, lam=None):
        """The same sum as _theta_vjp in HIP (csrc/kfield.hip: nksr_kernel_rows_vjp + nksr_voxel_psi_vjp): the per-row factors
        from two field evaluations with the rows' support (u = R' alpha, v = R' lambda are f / grad f at the sites), then one
        thread per (site, level) recomputes its row's forward and pushes the cotangents into the basis features (trilinear
        stencil), the neighbours' psi and the interpolator weights; psi_j = f_j + MLP(f_j) is taken back per voxel.  Returns
        the gradients in the order of ``_theta()``.  (kernel_rows_torch.py stays the reference the tests differentiate.)"""
        from .._lib import ThetaGradT
        dev, L, K = self.device, self.svh.depth, self.kdim
        al = alpha.detach().to(dev, torch.float32).contiguous()
        lm = lam.detach().to(dev, torch.float32).contiguous() if lam is not None else None
        gfeat = [torch.zeros_like(self._feat[d]) for d in range(L)]
        gpsi = [torch.zeros_like(self._feat[d]) for d in range(L)]
        gmlp = [torch.zeros_like(self._mlp[d]) for d in range(L)]
        tg = ThetaGradT()
        for d in range(L):
            tg.gfeat[d] = ptr(gfeat[d]) if gfeat[d].numel() else None
            tg.gpsi[d] = ptr(gpsi[d]) if gpsi[d].numel() else None
            tg.gmlp[d] = ptr(gmlp[d])
        with torch.no_grad():
            for xyz, grad_rows, sw, coeff in sets:
                if xyz is None or xyz.shape[0] == 0:
                    continue
                xs = xyz.detach().to(dev, torch.float32).contiguous()
                ra = self._evaluate_raw(al, xs, bool(grad_rows), active_only=True)
                u = (ra.gradient if grad_rows else ra.value) * float(sw)
                v = None
                if lm is not None:
                    rl = self._evaluate_raw(lm, xs, bool(grad_rows), active_only=True)
                    v = (rl.gradient if grad_rows else rl.value) * float(sw)
                a, b = coeff(u, v)
                ca = a.to(dev, torch.float32).contiguous() if (a is not None and lm is not None) else None
                cb = b.to(dev, torch.float32).contiguous() if b is not None else None
                if ca is None and cb is None:
                    continue
                call('nksr_kernel_rows_vjp', C.byref(self._hier), ptr(xs), xs.shape[0], int(bool(grad_rows)), int(self.approx_kernel_grad), float(sw),
                     ptr(ca), ptr(cb), ptr(al), ptr(lm) if ca is not None else None, C.byref(tg), stream())
            for d in range(L):
                n_d = self._feat[d].shape[0]
                if n_d:
                    call('nksr_voxel_psi_vjp', ptr(self._feat[d]), n_d, K, self.hidden, ptr(self._mlp[d]), ptr(gpsi[d]), ptr(gfeat[d]), ptr(gmlp[d]), stream())
        out = [gfeat[d].to(f.device, f.dtype) for d, f in enumerate(self._feat_in) if torch.is_tensor(f) and f.requires_grad]
        H = self.hidden
        sizes = [H * K, H, H * H, H, K * H, K]
        for d, m in enumerate(self._interps_in):
            if isinstance(m, torch.nn.Module):
                parts = dict(zip(('W1', 'b1', 'W2', 'b2', 'W3', 'b3'), torch.split(gmlp[d], sizes)))
                for name, q in m.named_parameters():
                    if q.requires_grad:
                        out.append(parts[name].reshape(q.shape).to(q.device, q.dtype))
        return out

    def solve(self, pos_xyz, normal_xyz, normal_value, pos_weight, normal_weight, reg_weight=1.0, fused_mode=True,
              pos_sorted_keys=None, normal_sorted_keys=None, segments=None):
        """``fused_mode=True`` (the reference's memory-lean operator, examples/recons_waymo.py:33): matrix-free solve,
        no assembly; ``False``: assemble the CSR and stream it (solve_non_fused -- the path the training code needs,
        models/nksr_net.py:105-112).  DESIGN.md section 3.5 has the cost model of the two."""
        if fused_mode:
            return self.solve_fused(pos_xyz, normal_xyz, normal_value, pos_weight, normal_weight, reg_weight,
                                    pos_sorted_keys, normal_sorted_keys, segments=segments)
        if segments is not None and segments.nseg > 1:
            raise RuntimeError('batched chunk solves run through the matrix-free solve (fused_mode=True)')
        return self.solve_non_fused(pos_xyz, normal_xyz, normal_value, pos_weight, normal_weight, reg_weight,
                                    pos_sorted_keys, normal_sorted_keys)

    # ---- evaluation -------------------------------------------------------------------------------------
    def _evaluate_f_model(self, xyz, grad, max_points=1 << 22):
        if torch.is_grad_enabled() and (self.alpha.requires_grad or self._theta()):
            f, g = _EvaluateFunction.apply(self, self.alpha, xyz.detach(), bool(grad), max_points, *self._theta())
            return EvaluationResult(f, g if grad else None)
        return self._evaluate_raw(self.alpha, xyz, grad, max_points)

    def _alpha_hier(self, alpha):
        """A copy of the hierarchy whose psi arrays hold alpha_j psi_j: evaluation then gathers ONE 16-byte value per neighbour
        (these per-point kernels are bound by the number of gather instructions).  Rebuilt when alpha changes."""
        key = (alpha.data_ptr(), alpha._version, str(self.device))
        if self._apsi_key != key:
            off = self.svh.offsets
            self._apsi = [(self._psi[d] * alpha[off[d]:off[d] + self._psi[d].shape[0], None]).contiguous() for d in range(self.svh.depth)]
            h = HierT.from_buffer_copy(self._hier)
            for d in range(self.svh.depth):
                h.lv[d].psi = ptr(self._apsi[d])
            self._apsi_hier, self._apsi_key = h, key
        return self._apsi_hier

    def _evaluate_raw(self, alpha, xyz, grad, max_points=1 << 22, active_only=False):
        n = xyz.shape[0]
        xyz = xyz.to(self.device)
        alpha = alpha.detach().contiguous()
        if alpha is self.alpha or alpha.data_ptr() == self.alpha.data_ptr():
            hier, alpha_arg = self._alpha_hier(alpha), None
        else:
            hier, alpha_arg = self._hier, alpha
        f = torch.empty(n, dtype=torch.float32, device=self.device)
        g = torch.empty((n, 3), dtype=torch.float32, device=self.device) if grad else None
        for s in range(0, n, max_points):
            e = min(n, s + max_points)
            xs = xyz[s:e].contiguous()
            fs = f[s:e]
            gs = g[s:e] if grad else None
            call('nksr_evaluate_f', C.byref(hier), ptr(alpha_arg), ptr(xs), e - s, int(self.approx_kernel_grad), int(bool(active_only)),
                 ptr(fs), ptr(gs), stream())
        return EvaluationResult(f, g)

    def to_(self, device):
        device = torch.device(device)
        self.svh.to_(device)
        self._feat = [t.to(device) for t in self._feat]
        self._psi = [t.to(device) for t in self._psi]
        self._mlp = [t.to(device) for t in self._mlp]
        self.alpha = self.alpha.to(device)           # (the setter drops the evaluation cache: it points into the old arrays)
        self.matrix = None
        self._fused_op = self._pc = None
        if device.type == 'cuda':
            self._hier = self._make_hier()
        if self.mask_field is not None:
            self.mask_field.to_(device)
        return self


class _SolveFunction(torch.autograd.Function):
    """alpha(normal targets, theta) for the system of the field's last solve:  A(theta) alpha = b(theta, n),
    A = sum_r R'_r^T R'_r + reg I,  b = sum_r R'_r t'_r  (R' = sqrt(w) R, t' = sqrt(w) n on the gradient rows, 0 on the position rows).
    With A lambda = dL/dalpha:  dL/dn = w_n Q lambda  ((Q lambda)[k, a] is d/dx_a of the kernel field with coefficients lambda at
    normal site k -- one PCG solve and one gradient evaluation) and
    dL/dtheta = sum_r dR'_r . [(t'_r - u_r) lambda - v_r alpha],  u = R' alpha, v = R' lambda  (KernelField._theta_vjp)."""

    @staticmethod
    def forward(ctx, field, alpha, pos_xyz, normal_xyz, normal_value, pos_weight, normal_weight, *theta):
        ctx.field, ctx.pos_xyz, ctx.normal_xyz = field, pos_xyz, normal_xyz
        ctx.pos_weight, ctx.normal_weight, ctx.n_theta = pos_weight, normal_weight, len(theta)
        ctx.normal_value = normal_value.detach()
        return alpha.clone()

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_alpha):
        fld = ctx.field
        lam = fld._solve_system(g_alpha.to(torch.float32))
        g_n = None
        if ctx.normal_xyz is not None and ctx.normal_value.numel():
            g_n = ctx.normal_weight * fld._evaluate_raw(lam, ctx.normal_xyz, True).gradient
        g_theta = []
        if ctx.n_theta:
            alpha = fld.alpha.detach()
            swp, swn = ctx.pos_weight ** 0.5, ctx.normal_weight ** 0.5
            tn = ctx.normal_value.to(fld.device, torch.float32) * swn if ctx.normal_value.numel() else None
            sets = [(ctx.pos_xyz, False, swp, lambda u, v: (-u, -v)),
                    (ctx.normal_xyz, True, swn, lambda u, v: ((tn - u) if tn is not None else -u, -v))]
            g_theta = fld._theta_vjp(sets, alpha, lam)
        return (None, None, None, None, g_n, None, None) + tuple(g_theta)


class _EvaluateFunction(torch.autograd.Function):
    """f(x) and grad f(x) as functions of alpha (linear) and theta: dL/dalpha = G_x^T g_f + Q_x^T g_grad, the set-up pass of the
    matrix-free operator over the kernel rows of the query points; dL/dtheta = sum_x dR_x . (g alpha) (KernelField._theta_vjp).
    Query points are not differentiated.  Support: the kernel rows exist only where the query lies in an active cell of the
    level, so the FORWARD of this (training) path is evaluated with the same support (nksr_evaluate_f active_only) -- the
    inference path (no autograd) also adds the levels whose neighbours a query outside every active cell still touches."""

    @staticmethod
    def forward(ctx, field, alpha, xyz, want_grad, max_points, *theta):
        ctx.field, ctx.xyz, ctx.want_grad, ctx.n_theta = field, xyz, want_grad, len(theta)
        ctx.alpha = alpha.detach()
        res = field._evaluate_raw(alpha, xyz, want_grad, max_points, active_only=True)
        g = res.gradient if want_grad else torch.zeros((0, 3), dtype=torch.float32, device=res.value.device)
        return res.value, g

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_f, g_grad):
        fld = ctx.field
        if ctx.xyz.shape[0] == 0:           # no queries: zero gradients
            th = fld._theta()[:ctx.n_theta]
            return (None, torch.zeros_like(ctx.alpha), None, None, None) + tuple(torch.zeros_like(q) for q in th)
        use_g = ctx.want_grad and g_grad is not None and g_grad.numel() > 0
        g_f = g_f if g_f is not None else torch.zeros(ctx.xyz.shape[0], device=fld.device)
        op = fld.fused_operator(ctx.xyz, ctx.xyz if use_g else None, g_grad if use_g else None, 1.0, 1.0, pos_value=g_f)
        b, _ = fld.fused_rhs_diag(op, 0.0)
        g_theta = []
        if ctx.n_theta:
            sets = [(ctx.xyz, False, 1.0, lambda u, v: (None, g_f.to(torch.float32)))]
            if use_g:
                sets.append((ctx.xyz, True, 1.0, lambda u, v: (None, g_grad.to(torch.float32))))
            g_theta = fld._theta_vjp(sets, ctx.alpha, None)
        return (None, b, None, None, None) + tuple(g_theta)


class _PackedInterpolator:
    """Interpolator stand-in carrying only the packed weights (used when a field is re-created from
    a payload: serialisation, chunk exchange)."""

    def __init__(self, kernel_dim, hidden_dim, packed):
        self.kernel_dim, self.hidden_dim, self._packed = int(kernel_dim), int(hidden_dim), packed

    def packed(self):
        return self._packed


def save_field(field, path):
    """Serialise a solved KernelField (hierarchy keys, basis features, alpha, interpolator weights,
    global scale).  SURVEY.md section 8(f)-3: the reference has no on-disk format for solved fields;
    this enables spill-to-disk of chunks and checkpointing."""
    from ..chunking import pack_field
    ints, flts = pack_field(field)
    torch.save({'format': 'nksr_amd.KernelField.v2', 'adaptive_depth': int(getattr(field.mask_field, 'adaptive_depth', 1)), 'ints': ints.cpu(), 'flts': flts.cpu(), 'voxel_size': field.svh.voxel_size,
                'hidden': field.hidden, 'kdim': field.kdim, 'scale': field.scale, 'mlp': [m.cpu() for m in field._mlp],
                'solve_info': field.solve_info}, path)


def load_field(path, device):
    from ..chunking import unpack_field
    from .mask_fields import LayerField
    st = torch.load(path, map_location='cpu')
    if st.get('format') != 'nksr_amd.KernelField.v2':
        raise RuntimeError('%s is not a serialised KernelField' % path)
    interps = [_PackedInterpolator(st['kdim'], st['hidden'], m) for m in st['mlp']]
    fld = unpack_field(st['ints'], st['flts'], st['voxel_size'], interps, torch.device(device))
    fld.set_scale(st['scale'])
    fld.solve_info = st.get('solve_info', {})
    if fld.mask_field is None:          # a UDF (NeuralField) mask travels inside the payload
        fld.set_mask_field(LayerField(fld.svh, st.get('adaptive_depth', 1)))
    return fld
