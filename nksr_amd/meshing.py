"""Dual marching cubes + MISE -- host orchestration of csrc/meshing.hip.

Mirrors ``field.extract_dual_mesh(mise_iter=0, grid_upsample=1, max_points=-1)`` of the
reference (call sites examples/recons_simple.py:27, recons_scannet.py:29,
recons_colored_mesh.py:30, models/nksr_net.py:214,284) returning ``.v [V,3] f32``,
``.f [T,3] int``, ``.c [V,3]`` (NKSR-USAGE.md:52,79).  Steps:
base dual cells -> (lattice vertices, f) -> MISE split of sign-changing cells -> 256-case
table -> edge-keyed vertex dedup (radix sort + unique) -> mask trim -> world units.
Host syncs happen only where a size (cells, vertices, triangles) must reach the host.
"""
import os

import torch

from . import ops
from ._lib import call, ptr, stream
from .fields.base_field import MeshingResult


def _cell_vertices(cell_keys_raw):
    """sorted-unique cells, their sorted-unique lattice vertices, and the [ncell,8] corner table"""
    dev = cell_keys_raw.device
    cells = ops.sort_unique(cell_keys_raw, maybe_sorted=True)      # base cells of Morton-ordered voxels / children of sorted cells
    nc = cells.numel()
    ck = torch.empty(nc * 8, dtype=torch.int64, device=dev)
    call('nksr_cell_corner_keys', ptr(cells), nc, ptr(ck), stream())
    vkeys = ops.sort_unique(ops.dedup_corner_keys(cells) if nc >= (1 << 16) else ck)
    vhash = ops.HashTable(vkeys)      # key -> index into vkeys: one or two probes instead of a 21-step binary search per lookup
    cidx = vhash.query(ck).view(nc, 8)
    return cells, vkeys, cidx, vhash


def extract_dual_mesh(field, mise_iter=0, grid_upsample=1, max_points=-1):
    # field.dual_graph = 'adaptive': cells as large as the hierarchy level that carries them (the reference's dual graph of the
    # flattened levels); 'lattice' (default): one uniform lattice over the adaptive support.
    # A chunked field meshes its union hierarchy (levels < adaptive_depth on the global lattice) with the blended field like any
    # other.  Spread over several ranks, every rank meshes the hexahedra around the octree corners inside its own cores (the leaves
    # one halo deep across the seam are its neighbours' -- chunking.halo_inner) and rank 0 merges the pieces by the vertices'
    # (size, key) pair names (dist.merge_named).
    spread = hasattr(field, 'finalize_mesh') and (getattr(field, 'world_size', 1) > 1 or getattr(field, 'distributed', False))
    if getattr(field, 'dual_graph', 'lattice') == 'adaptive':
        if not spread:
            return _extract_adaptive(field, mise_iter, grid_upsample, max_points)
        from . import chunking
        need = chunking.halo_inner(field.svh.voxel_size, getattr(field, 'meshing_depth', 1), 'adaptive')
        if getattr(field, 'world_size', 1) > 1 and (getattr(field, 'halo_inner', None) is None or field.halo_inner < need - 1e-6 * need):
            # (never a mesh from halos that are too thin for it: the field was reconstructed for the lattice mesher)
            raise RuntimeError("dual_graph='adaptive' on a chunked field spread over several ranks needs halos %.3g deep (set "
                               "Reconstructor.dual_graph = 'adaptive' BEFORE reconstruct(); this field's are %s)"
                               % (need, getattr(field, 'halo_inner', None)))
        return field.finalize_mesh_named(_extract_adaptive(field, mise_iter, grid_upsample, max_points, owned=True))
    res = _extract(field, mise_iter, grid_upsample, max_points)
    if hasattr(field, 'finalize_mesh'):       # distributed fields gather + stitch the pieces (collective)
        res = field.finalize_mesh(res)
    return res


def _extract(field, mise_iter, grid_upsample, max_points):
    svh = field.svh
    dev = svh.device
    g0 = svh.level(0)
    w0 = svh.voxel_size
    U = int(grid_upsample)
    empty = MeshingResult(torch.zeros((0, 3), dtype=torch.float32, device=dev), torch.zeros((0, 3), dtype=torch.int64, device=dev))
    empty.edge_vkey = torch.zeros(0, dtype=torch.int64, device=dev)
    empty.edge_axis = torch.zeros(0, dtype=torch.int8, device=dev)
    batch = max_points if (max_points is not None and max_points > 0) else (1 << 22)
    if U < 1 or mise_iter < 0:
        raise RuntimeError('grid_upsample must be >= 1 and mise_iter >= 0')
    owned_only = hasattr(field, 'base_cell_mask') and getattr(field, 'world_size', 1) > 1
    # a rank's piece says which of its vertices another rank may emit too (rank 0 groups only those); every rank decides alike --
    # the gather's tensor count is part of the collective -- and an empty piece carries empty flags
    use_seam = owned_only and dev.type == 'cuda' and hasattr(field, 'seam_flags') and os.environ.get('NKSR_SEAM_FLAGS', '1') != '0'
    if use_seam:
        empty.seam_flag = torch.zeros(0, dtype=torch.uint8, device=dev)
    # levels whose dual cells are meshed: the finest, and -- LayerField(dec_svh, adaptive_depth), models/nksr_net.py:132 -- the coarser
    # ones below adaptive_depth, which cover what the finest level leaves open (a structure head that stops at level 1, or input
    # sparser than the finest voxels): their extent at the SAME lattice resolution -- one uniform lattice over the adaptive
    # support, so there are no level transitions to stitch
    adaptive = max(1, min(int(getattr(field, 'meshing_depth', 1)), svh.depth))
    levels = [d for d in range(adaptive) if svh.level(d).num_voxels > 0]
    if not levels:
        return empty
    # lattice / cell keys are 21-bit-per-axis Morton codes biased by 2^20 (csrc/meshing.hip): the refined lattice
    # coordinate ijk * U * 2^mise_iter (+ one cell) must stay inside, or keys would wrap silently
    reach = max(((int(svh.level(d).ijk.abs().max()) + 4) << d) for d in levels) * U * (1 << int(mise_iter)) + 2
    if reach >= (1 << 20):
        raise RuntimeError('mesh lattice out of range: |ijk| * grid_upsample * 2^mise_iter = %d >= 2^20; recentre the cloud '
                           '(or lower mise_iter / grid_upsample)' % reach)

    raw = torch.empty(0, dtype=torch.int64, device=dev)
    if g0.num_voxels:
        flags = torch.empty(g0.num_voxels, dtype=torch.int32, device=dev)
        call('nksr_base_cell_flags', ptr(g0.nbr), g0.num_voxels, ptr(flags), stream())
        if owned_only:      # distributed fields: cells this rank owns + a one-cell halo (evaluated, not meshed)
            flags = (flags * field.base_cell_halo_mask(g0.ijk).to(torch.int32)).contiguous()
        sel = ops.compact(flags)
        raw = torch.empty(sel.numel() * U ** 3, dtype=torch.int64, device=dev)
        if sel.numel():
            call('nksr_base_cell_keys', ptr(g0.ijk), ptr(sel), sel.numel(), U, ptr(raw), stream())
    for d in levels:
        if d == 0:
            continue
        gd = svh.level(d)
        fl = torch.empty(gd.num_voxels, dtype=torch.int32, device=dev)
        call('nksr_base_cell_flags', ptr(gd.nbr), gd.num_voxels, ptr(fl), stream())
        sd = ops.compact(fl)
        if sd.numel():
            S = U << d
            rd = torch.empty(sd.numel() * S ** 3, dtype=torch.int64, device=dev)
            call('nksr_level_cell_keys', ptr(gd.ijk), ptr(sd), sd.numel(), d, U, ptr(rd), stream())
            if owned_only:  # the same ownership rule, on the finest voxel that contains the lattice cell
                gc = torch.empty((rd.numel(), 3), dtype=torch.int32, device=dev)
                call('nksr_decode_keys', ptr(rd), rd.numel(), -1, ptr(gc), stream())
                rd = rd[field.base_cell_halo_mask(torch.div(gc, U, rounding_mode='floor').to(torch.int32))]
            raw = torch.cat([raw, rd])
    if raw.numel() == 0:
        return empty

    h = w0 / U
    prev = None          # (vertex keys, values, active cell keys) of the coarser MISE level
    for m in range(mise_iter + 1):
        cells, vkeys, cidx, vhash = _cell_vertices(raw)
        nv, nc = vkeys.numel(), cells.numel()
        pos = torch.empty((nv, 3), dtype=torch.float32, device=dev)
        call('nksr_lattice_positions', ptr(vkeys), nv, float(h), float(0.5 * w0), ptr(pos), stream())
        f = field._evaluate_f_model(pos, False, max_points=batch).value
        if prev is not None:   # hanging vertices take the coarse interpolant: no T-junction cracks
            ch, ah = prev[0], prev[2]
            call('nksr_mise_constrain', ptr(vkeys), nv, ptr(f), ptr(ch.hkeys), ptr(ch.hvals), ch.cap, ptr(prev[1]), ptr(ah.hkeys), ptr(ah.hvals),
                 ah.cap, stream())
        config = torch.empty(nc, dtype=torch.int32, device=dev)
        ntri = torch.empty(nc + 1, dtype=torch.int32, device=dev)
        ntri[nc] = 0
        call('nksr_cell_config', ptr(cidx), ptr(f), nc, ptr(config), ptr(ntri), stream())
        if m < mise_iter:
            act = torch.empty(nc, dtype=torch.int32, device=dev)
            call('nksr_cell_active_flags', ptr(config), nc, ptr(act), stream())
            asel = ops.compact(act)
            if asel.numel() == 0:
                return empty
            raw = torch.empty(asel.numel() * 8, dtype=torch.int64, device=dev)
            call('nksr_cell_children', ptr(cells), ptr(asel), asel.numel(), ptr(raw), stream())
            prev = (vhash, f, ops.HashTable(cells[asel.long()].contiguous()))      # coarse vertices / refined coarse cells (sorted subset)
            h = h / 2

    if owned_only:      # halo cells emit nothing: ownership follows the base voxel that contains the cell
        gc = torch.empty((nc, 3), dtype=torch.int32, device=dev)
        call('nksr_decode_keys', ptr(cells), nc, -1, ptr(gc), stream())
        base_ijk = torch.div(gc, U * (1 << mise_iter), rounding_mode='floor').to(torch.int32)
        keep_c = field.base_cell_mask(base_ijk).to(torch.int32)
        ntri[:nc] *= keep_c
        config = (config * keep_c).contiguous()     # configuration 0 emits nothing in nksr_mc_emit
    tri_off = ops.exclusive_sum_i32(ntri)
    T = int(tri_off[nc].item())
    if T == 0:
        return empty
    ekeys = torch.empty(T * 3, dtype=torch.int64, device=dev)
    call('nksr_mc_emit', ptr(cidx), ptr(config), ptr(tri_off), nc, ptr(ekeys), stream())
    uek = ops.sort_unique(ops.dedup_keys(ekeys) if ekeys.numel() >= (1 << 19) else ekeys)      # every edge vertex is named by ~6 triangle corners
    faces = ops.HashTable(uek).query(ekeys).view(T, 3)
    ne = uek.numel()
    verts = torch.empty((ne, 3), dtype=torch.float32, device=dev)
    call('nksr_mc_vertices', ptr(uek), ne, ptr(vkeys), ptr(vhash.hkeys), ptr(vhash.hvals), vhash.cap, ptr(pos), ptr(f), float(h), ptr(verts), stream())

    # canonical identity of every mesh vertex: (lattice key of the lower end point, axis)
    ev = torch.div(uek, 3, rounding_mode='floor')
    edge_vkey, edge_axis = vkeys[ev], (uek - ev * 3).to(torch.int8)
    verts, faces, (edge_vkey, edge_axis) = _trim(field, verts, faces, (edge_vkey, edge_axis), masked=True)
    res = _result(field, verts, faces)
    res.edge_vkey, res.edge_axis, res.lattice_h = edge_vkey, edge_axis, h
    if use_seam:
        res.seam_flag = field.seam_flags(edge_vkey, edge_axis, U * (1 << mise_iter))       # which vertices rank 0 has to group
    return res


def _trim(field, verts, faces, per_vertex=(), masked=True, drop_unused=False):
    """Mask trim (triangles with a vertex outside the mask field go) and removal of the vertices no triangle uses."""
    dev = verts.device
    ne = verts.shape[0]
    keep_v = field.mask_vertices(verts) if masked else None
    trimmed = keep_v is not None and not bool(keep_v.all())
    if trimmed:
        faces = faces[keep_v[faces.long()].all(1)]
    if trimmed or drop_unused:
        used = torch.zeros(ne, dtype=torch.int32, device=dev)
        used[faces.reshape(-1).long()] = 1
        remap = ops.exclusive_sum_i32(used)
        vsel = ops.compact(used).long()
        verts = verts[vsel]
        per_vertex = tuple(a[vsel] for a in per_vertex)
        faces = remap[faces.long()]
    return verts, faces, per_vertex


def _result(field, verts, faces):
    v_world = verts / field.scale if field.scale != 1.0 else verts
    colors = None
    if field.texture_field is not None:
        colors = field.texture_field.evaluate_color(v_world)
    return MeshingResult(v_world, faces.long(), colors)


# ---- marching cubes on the adaptive dual graph (specification: oracle/dual_adaptive.py) ------------------------------------------------
def _leaf_coordinates(svh, adaptive):
    """Per level d < adaptive: integer coordinates [m_d, 3] of the octree's leaves -- level-0 voxels, voxels without children, and
    the children a refined voxel does not have ("virtual": empty space next to a refined region still carries a sample)."""
    dev = svh.device
    out = [[] for _ in range(adaptive)]
    for d in range(adaptive):
        g = svh.level(d)
        if g.num_voxels == 0:
            continue
        if d == 0:
            out[0].append(g.ijk)
            continue
        child = svh.level(d - 1)
        internal = torch.zeros(g.num_voxels, dtype=torch.bool, device=dev)
        if child.num_voxels:
            pi = g.hash.query((child.keys >> 3).contiguous())             # the key of a parent is its child's without the low octant
            internal[pi[pi >= 0].long()] = True
        out[d].append(g.ijk[~internal])
        ik = g.keys[internal]
        if ik.numel():
            ck = ((ik[:, None] << 3) | torch.arange(8, dtype=torch.int64, device=dev)[None]).reshape(-1).contiguous()
            vk = ck[child.hash.query(ck) < 0].contiguous()
            if vk.numel():
                vijk = torch.empty((vk.numel(), 3), dtype=torch.int32, device=dev)
                call('nksr_decode_keys', ptr(vk), vk.numel(), d - 1, ptr(vijk), stream())
                out[d - 1].append(vijk)
    return [torch.cat(o) if o else torch.zeros((0, 3), dtype=torch.int32, device=dev) for o in out]


class _CellTable:
    """The primal cells of all sizes as one table, smallest first, each size sorted by key: id = offset + rank (csrc/meshing.hip
    nksr_cell_table_t).  ``pieces[lam]`` = list of (sorted keys, values or None = not evaluated yet); a size that comes as ONE piece
    is in order already, several pieces (kept cells + the children of split ones) are merged by a radix sort.  ``f`` rides along
    (NaN = not evaluated yet; ``all_new`` / ``any_new`` say so on the host)."""

    def __init__(self, pieces, dev):
        from ._lib import CELL_SIZES, CellTableT
        self.lams = sorted(l for l in pieces if sum(p[0].numel() for p in pieces[l]))
        if len(self.lams) > CELL_SIZES:
            raise RuntimeError('adaptive dual graph: more than %d cell sizes' % CELL_SIZES)
        self.keys, self.hashes, self.offset = {}, {}, {}
        t = CellTableT()
        n, fs, ks, ls = 0, [], [], []
        self.all_new = all(p[1] is None for l in self.lams for p in pieces[l] if p[0].numel())
        self.any_new = any(p[1] is None for l in self.lams for p in pieces[l] if p[0].numel())
        nan = float('nan')
        for i, l in enumerate(self.lams):
            ps = [p for p in pieces[l] if p[0].numel()]
            if len(ps) == 1:
                ko = ps[0][0].contiguous()
                f = ps[0][1] if ps[0][1] is not None else (None if self.all_new else torch.full((ko.numel(),), nan, dtype=torch.float32, device=dev))
            else:
                k = torch.cat([p[0] for p in ps])
                ko, order = ops.sort_pairs(k, torch.arange(k.numel(), dtype=torch.int32, device=dev))
                f = None
                if not self.all_new:
                    f = torch.cat([p[1] if p[1] is not None else torch.full((p[0].numel(),), nan, dtype=torch.float32, device=dev) for p in ps])[order.long()]
            self.keys[l], self.hashes[l], self.offset[l] = ko, ops.HashTable(ko), n
            t.lam[i], t.offset[i], t.hcap[i], t.hkeys[i], t.hvals[i] = l, n, self.hashes[l].cap, ptr(self.hashes[l].hkeys), ptr(self.hashes[l].hvals)
            fs.append(f)
            ks.append(ko)
            ls.append(torch.full((ko.numel(),), l, dtype=torch.int32, device=dev))
            n += ko.numel()
        t.nlev = len(self.lams)
        self.struct, self.n = t, n
        if n >= (1 << 30):
            raise RuntimeError('adaptive dual graph: %d cells (vertex names hold 30-bit cell ids)' % n)
        one = len(ks) == 1
        self.key = (ks[0] if one else torch.cat(ks)) if ks else torch.zeros(0, dtype=torch.int64, device=dev)
        self.lam = (ls[0] if one else torch.cat(ls)) if ls else torch.zeros(0, dtype=torch.int32, device=dev)
        self.f = None if self.all_new else ((fs[0] if one else torch.cat(fs)) if fs else torch.zeros(0, dtype=torch.float32, device=dev))


def _names5(uek, tab):
    """Local vertex names (cell id A 2^33 + axis 2^31 + cell id B) -> [n, 5] (size A, key A, axis, size B, key B): what the
    ids stand for, the same on every rank; ids ascend with (size, key), so the lexicographic order of the rows is the order of uek."""
    a, b = (uek >> 33).long(), (uek & 0x7FFFFFFF).long()
    return torch.stack([tab.lam[a].to(torch.int64), tab.key[a], (uek >> 31) & 3, tab.lam[b].to(torch.int64), tab.key[b]], 1).contiguous()


def _extract_adaptive(field, mise_iter, grid_upsample, max_points, level_ijk=None, owned=False):
    """``level_ijk`` (tests): the voxel coordinates per level instead of the field's hierarchy.  ``owned`` (a field spread over
    ranks): only the cells within the halo depth of this rank's cores enter (the field is exact there and nowhere else), every
    dual cell is configured and takes part in the MISE splits, and the hexahedra around the corners inside this rank's cores emit.
    Host syncs: one per size that must reach the host (unique corners, active cells, split / kept cells per size, triangles, vertices)."""
    svh = field.svh
    dev = svh.device
    w0 = svh.voxel_size
    U, M = int(grid_upsample), int(mise_iter)
    if U < 1 or U > 8 or M < 0:
        raise RuntimeError('grid_upsample must be in [1, 8] and mise_iter >= 0')
    empty = MeshingResult(torch.zeros((0, 3), dtype=torch.float32, device=dev), torch.zeros((0, 3), dtype=torch.int64, device=dev))
    if owned:
        empty.vertex_names5 = torch.zeros((0, 5), dtype=torch.int64, device=dev)
    batch = max_points if (max_points is not None and max_points > 0) else (1 << 22)
    if level_ijk is None:
        adaptive = max(1, min(int(getattr(field, 'meshing_depth', 1)), svh.depth))
        leaf = _leaf_coordinates(svh, adaptive)
    else:
        leaf = _leaf_coordinates(_LevelsOnly(level_ijk, w0, dev), len(level_ijk))
    if not any(c.numel() for c in leaf):
        return empty
    reach = max(((int(c.abs().max()) + 2) << d) for d, c in enumerate(leaf) if c.numel()) * U * (1 << M) + 2
    if reach >= (1 << 20):
        raise RuntimeError('mesh lattice out of range: |ijk| * grid_upsample * 2^mise_iter = %d >= 2^20; recentre the cloud '
                           '(or lower mise_iter / grid_upsample)' % reach)
    u = w0 / U / (1 << M)
    sub = torch.tensor([[a, b, c] for a in range(U) for b in range(U) for c in range(U)], dtype=torch.int32, device=dev)
    pieces = {}
    for d, c in enumerate(leaf):
        if not c.numel():
            continue
        cc = c.contiguous() if U == 1 else (c[:, None, :] * U + sub[None]).reshape(-1, 3).contiguous()
        k = torch.empty(cc.shape[0], dtype=torch.int64, device=dev)
        call('nksr_encode_keys', ptr(cc), cc.shape[0], -1, ptr(k), stream())
        k = ops.sort_unique(k, maybe_sorted=(U == 1 and d == 0))       # (the finest voxels come in key order)
        if owned and k.numel():
            cp = torch.empty((k.numel(), 3), dtype=torch.float32, device=dev)
            call('nksr_adaptive_positions', ptr(k), ptr(torch.full((k.numel(),), d + M, dtype=torch.int32, device=dev)), k.numel(), float(u), ptr(cp), stream())
            k = k[field.near_owned(cp, float(field.halo_inner or 0.0) - 0.5 * w0)].contiguous()      # (one rank: everything)
        if k.numel():
            pieces[d + M] = [(k, None)]
    if not pieces:
        return empty
    for m in range(M + 1):
        tab = _CellTable(pieces, dev)
        pos = torch.empty((tab.n, 3), dtype=torch.float32, device=dev)
        call('nksr_adaptive_positions', ptr(tab.key), ptr(tab.lam), tab.n, float(u), ptr(pos), stream())
        if tab.all_new:
            f = field._evaluate_f_model(pos, False, max_points=batch).value.contiguous()
        else:
            f = tab.f
            if tab.any_new:
                sel = torch.nonzero(torch.isnan(f)).flatten()
                f[sel] = field._evaluate_f_model(pos[sel].contiguous(), False, max_points=batch).value
        # dual cells: the corners of all cells, sorted by the key of k - 1; the eight cells around each (a corner that lacks one
        # keeps its -1: configuration 0, nothing emitted, csrc/meshing.hip k_cell_config)
        ck = torch.empty(tab.n * 8, dtype=torch.int64, device=dev)
        for l in tab.lams:
            call('nksr_adaptive_corner_keys', ptr(tab.keys[l]), tab.keys[l].numel(), l, ptr(ck[tab.offset[l] * 8:]), stream())
        ckeys = ops.sort_unique(ops.dedup_keys(ck) if ck.numel() >= (1 << 19) else ck)      # (a corner is named by up to eight cells)
        nc = ckeys.numel()
        cidx = torch.empty((nc, 8), dtype=torch.int32, device=dev)
        call('nksr_adaptive_dual_cells', ptr(ckeys), nc, tab.struct, ptr(cidx), stream())
        config = torch.empty(nc, dtype=torch.int32, device=dev)
        ntri = torch.empty(nc + 1, dtype=torch.int32, device=dev)
        ntri[nc] = 0
        call('nksr_cell_config', ptr(cidx), ptr(f), nc, ptr(config), ptr(ntri), stream())
        if m < M:   # MISE: every cell around a sign-changing dual cell is split into 8; the rest keep their values
            act = torch.empty(nc, dtype=torch.int32, device=dev)
            call('nksr_cell_active_flags', ptr(config), nc, ptr(act), stream())
            asel = ops.compact(act)
            if asel.numel() == 0:
                return empty
            split = torch.zeros(tab.n, dtype=torch.int32, device=dev)
            split[cidx[asel.long()].reshape(-1).long()] = 1
            pieces = {}
            for l in tab.lams:
                lo, n_l = tab.offset[l], tab.keys[l].numel()
                fl = f[lo:lo + n_l]
                if l == 0:                                                   # (the finest unit is not split)
                    pieces.setdefault(l, []).append((tab.keys[l], fl))
                    continue
                sp = split[lo:lo + n_l].contiguous()
                ssel = ops.compact(sp)
                if ssel.numel() < n_l:
                    ksel = (ops.compact(1 - sp) if ssel.numel() else torch.arange(n_l, dtype=torch.int32, device=dev)).long()
                    pieces.setdefault(l, []).append((tab.keys[l][ksel], fl[ksel]))
                if ssel.numel():
                    ch = torch.empty(ssel.numel() * 8, dtype=torch.int64, device=dev)
                    call('nksr_cell_children', ptr(tab.keys[l]), ptr(ssel), ssel.numel(), ptr(ch), stream())
                    pieces.setdefault(l - 1, []).append((ops.sort_keys(ch), None))        # (nksr_cell_children writes corner order, not key order)
    if owned:           # the hexahedron of corner k (its key is that of k - 1) emits on the rank whose core holds k
        kc = torch.empty((nc, 3), dtype=torch.int32, device=dev)
        call('nksr_decode_keys', ptr(ckeys), nc, -1, ptr(kc), stream())
        keep_c = field.owns_points((kc + 1).to(torch.float32) * float(u)).to(torch.int32)
        ntri[:nc] *= keep_c
        config = (config * keep_c).contiguous()     # configuration 0 emits nothing
    tri_off = ops.exclusive_sum_i32(ntri)
    T = int(tri_off[nc].item())
    if T == 0:
        return empty
    names = torch.empty(T * 3, dtype=torch.int64, device=dev)
    call('nksr_mc_emit_pairs', ptr(cidx), ptr(config), ptr(tri_off), nc, ptr(names), stream())
    names = names.view(T, 3)
    good = (names[:, 0] != names[:, 1]) & (names[:, 1] != names[:, 2]) & (names[:, 0] != names[:, 2])     # collapsed edges of degenerate cells
    names = names[good].contiguous().reshape(-1)
    if names.numel() == 0:
        return empty
    uek = ops.sort_unique(ops.dedup_keys(names) if names.numel() >= (1 << 19) else names)       # every vertex is named by ~6 triangle corners
    faces = ops.HashTable(uek).query(names).view(-1, 3)
    ne = uek.numel()
    verts = torch.empty((ne, 3), dtype=torch.float32, device=dev)
    call('nksr_pair_vertices', ptr(uek), ne, ptr(tab.key), ptr(tab.lam), ptr(pos), ptr(f), float(u), ptr(verts), stream())
    verts, faces, (uek,) = _trim(field, verts, faces, (uek,), masked=True)
    res = _result(field, verts, faces)
    res.vertex_name, res.cell_key, res.cell_lam, res.cell_f, res.fine_unit = uek, tab.key, tab.lam, f, u
    if owned:
        res.vertex_names5 = _names5(uek, tab)
    return res


class _LevelsOnly:
    """Voxel coordinates per level dressed as the part of a hierarchy _leaf_coordinates reads (tests: mixed-level patterns)."""

    def __init__(self, level_ijk, voxel_size, dev):
        from .svh import SparseGrid
        self.device, self.voxel_size, self.depth = dev, voxel_size, len(level_ijk)
        self._levels = []
        for d, ijk in enumerate(level_ijk):
            ijk = torch.as_tensor(ijk, dtype=torch.int32, device=dev).reshape(-1, 3).contiguous()
            k = torch.empty(ijk.shape[0], dtype=torch.int64, device=dev)
            call('nksr_encode_keys', ptr(ijk), ijk.shape[0], d, ptr(k), stream())
            self._levels.append(SparseGrid(ops.sort_unique(k), d, voxel_size))

    def level(self, d):
        return self._levels[d]
