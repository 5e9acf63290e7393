"""Oracle: dual marching cubes over the finest level + MISE refinement.

Restates field.extract_dual_mesh(mise_iter, grid_upsample, max_points) ->
mesh.v / mesh.f (examples/recons_simple.py:27, recons_scannet.py:29,
recons_colored_mesh.py:30, models/nksr_net.py:214,284; NKSR-USAGE.md:52,79).
Spec:
  * dual vertices = centres of active level-0 voxels; a base dual cell exists where
    the 8 mutually adjacent centres i + {0,1}^3 are all active
  * lattice at refinement m with upsample U: spacing h = w0 / (U 2^m),
    x = g*h + w0/2 for integer lattice coordinate g
  * MISE: cells whose 8 corner values do not share a sign are split into 8
  * final cells -> 256-case table (oracle.mc_tables); edge vertices are keyed by
    (lower lattice vertex, axis) and interpolated lower -> upper
"""
import numpy as np
from . import spec, mc_tables

REFINE_BIAS = 1 << 20


def lattice_key(g):
    b = g.astype(np.int64) + REFINE_BIAS
    assert (b >= 0).all() and (b < (1 << 21)).all(), "refined lattice out of range"
    k = spec._part1by2(b[..., 0]) | (spec._part1by2(b[..., 1]) << np.uint64(1)) | (spec._part1by2(b[..., 2]) << np.uint64(2))
    return k.astype(np.int64)


def lattice_decode(k):
    k = k.astype(np.uint64)
    g = np.stack([spec._compact1by2(k), spec._compact1by2(k >> np.uint64(1)), spec._compact1by2(k >> np.uint64(2))], -1)
    return (g.astype(np.int64) - REFINE_BIAS).astype(np.int32)


def lattice_positions(g, h, half_w0):
    return (g.astype(np.float32) * np.float32(h) + np.float32(half_w0)).astype(np.float32)


def base_cells(level0, upsample, coarser=()):
    """Lattice cells of the meshing domain: the dual cells of level 0 (8 mutually adjacent active centres) and -- for the
    levels 1 .. adaptive_depth-1 in ``coarser`` -- the extent of their dual cells at the same lattice resolution (the
    (U 2^d)^3 cells whose centre lies inside a level-d dual cell start at (2^d i + 2^(d-1) - 1) U)."""
    out = []
    for d, lev in [(0, level0)] + [(L.level, L) for L in coarser]:
        nbr = lev.nbr
        ok = np.ones(lev.n, bool)
        for co in spec.CORNER_OFFSETS:
            s = (co[0] + 1) * 9 + (co[1] + 1) * 3 + (co[2] + 1)
            ok &= nbr[:, s] >= 0
        base = lev.ijk[ok].astype(np.int64)
        S = upsample << d
        off = 0 if d == 0 else (1 << (d - 1)) - 1
        sub = np.array([[a, b, c] for a in range(S) for b in range(S) for c in range(S)], np.int64)
        out.append((((base << d) + off)[:, None, :] * upsample + sub[None]).reshape(-1, 3))
    return np.concatenate(out) if out else np.zeros((0, 3), np.int64)


def cell_vertices(cells):
    """Sorted-unique lattice vertices of a cell set and the [n,8] corner index table."""
    ck = np.unique(lattice_key(cells))
    cells = lattice_decode(ck).astype(np.int64)
    corners = (cells[:, None, :] + spec.CORNER_OFFSETS[None].astype(np.int64))
    vk = np.unique(lattice_key(corners.reshape(-1, 3)))
    idx = np.searchsorted(vk, lattice_key(corners.reshape(-1, 3))).reshape(-1, 8)
    return cells, vk, idx.astype(np.int32)


def _find(sorted_keys, q):
    pos = np.searchsorted(sorted_keys, q)
    pos_c = np.minimum(pos, max(len(sorted_keys) - 1, 0))
    hit = (pos < len(sorted_keys)) & (sorted_keys[pos_c] == q) if len(sorted_keys) else np.zeros(q.shape, bool)
    return np.where(hit, pos_c, -1)


def constrain_hanging(g, f, vk_coarse, f_coarse, active_keys):
    """MISE hanging-vertex rule: a refined vertex on a coarse edge / face takes
    the mean of the coarse end points / face corners unless every coarse cell sharing it was refined."""
    f = f.copy()
    g = g.astype(np.int64)
    odd = g & 1
    k = odd.sum(1)
    even = np.nonzero(k == 0)[0]          # coincide with coarse vertices: inherit their (constrained) value
    j = _find(vk_coarse, lattice_key(g[even] >> 1))
    f[even[j >= 0]] = f_coarse[j[j >= 0]]
    for idx in np.nonzero((k == 1) | (k == 2))[0]:
        gi, oi = g[idx], odd[idx]
        lo = [((gi[a] - 1) >> 1) if oi[a] else (gi[a] >> 1) - 1 for a in range(3)]
        cnt = [1 if oi[a] else 2 for a in range(3)]
        cells = np.array([[lo[0] + x, lo[1] + y, lo[2] + z] for x in range(cnt[0]) for y in range(cnt[1]) for z in range(cnt[2])])
        if (_find(active_keys, lattice_key(cells)) >= 0).all():
            continue
        vs = np.array([[(((gi[0] - 1) >> 1) + x) if oi[0] else gi[0] >> 1,
                        (((gi[1] - 1) >> 1) + y) if oi[1] else gi[1] >> 1,
                        (((gi[2] - 1) >> 1) + z) if oi[2] else gi[2] >> 1]
                       for x in range(oi[0] + 1) for y in range(oi[1] + 1) for z in range(oi[2] + 1)])
        j = _find(vk_coarse, lattice_key(vs))
        if (j >= 0).all():
            s = np.float32(0)
            for v in f_coarse[j]:
                s = np.float32(s + v)
            f[idx] = s * np.float32(0.5 if k[idx] == 1 else 0.25)
    return f


def extract(level0_voxel_size, level0, eval_fn, mise_iter=0, grid_upsample=1, mask_fn=None, info=None, coarser=()):
    """eval_fn(xyz[n,3] f32) -> f[n] f32;  mask_fn(xyz) -> bool[n] (True = keep).
    ``info`` (dict, optional) receives what the parity tests need to localise a topology difference:
    per MISE level the cell coordinates / corner table / (constrained) vertex values, and for the output
    mesh the cell of every triangle and the canonical identity (lattice key of the lower end point, axis)
    plus |f0 - f1| of every vertex."""
    w0 = float(level0_voxel_size)
    U = int(grid_upsample)
    cells = base_cells(level0, U, coarser)
    h = w0 / U
    prev = None
    for m in range(mise_iter + 1):
        cells, vk, cidx = cell_vertices(cells)
        g = lattice_decode(vk)
        pos = lattice_positions(g, h, 0.5 * w0)
        f = eval_fn(pos) if len(pos) else np.zeros(0, np.float32)
        f_raw = f
        if prev is not None and len(f):
            f = constrain_hanging(g, f, *prev)
        inside = f > 0
        ci = inside[cidx] if len(cidx) else np.zeros((0, 8), bool)
        config = (ci * (1 << np.arange(8))[None]).sum(1).astype(np.int32) if len(cidx) else np.zeros(0, np.int32)
        if info is not None:
            info.setdefault('levels', []).append({'cells': cells.copy(), 'cidx': cidx, 'f': f.copy(), 'f_raw': f_raw, 'vk': vk, 'pos': pos,
                                                  'config': config, 'h': h})
        if m < mise_iter:
            act = (config != 0) & (config != 255)
            prev = (vk, f, np.sort(lattice_key(cells[act])))
            cells = (cells[act][:, None, :] * 2 + spec.CORNER_OFFSETS[None].astype(np.int64)).reshape(-1, 3)
            h = h / 2
    # marching cubes on the final cell set
    ntri = mc_tables.TRI_COUNT[config]
    tri_edge_keys = []
    for t in range(mc_tables.TRI_TABLE.shape[1]):
        sel = np.nonzero(ntri > t)[0]
        e = mc_tables.TRI_TABLE[config[sel], t]                       # [k,3] edge ids
        lo = cidx[sel[:, None], mc_tables.EDGE_LO[e]]                 # lower lattice vertex index
        key = lo.astype(np.int64) * 3 + mc_tables.EDGE_AXIS[e]
        tri_edge_keys.append((sel, np.full(len(sel), t), key))
    if not tri_edge_keys or sum(len(s) for s, _, _ in tri_edge_keys) == 0:
        if info is not None:
            info.update({'tri_cell': np.zeros((0, 3), np.int64), 'vert_vkey': np.zeros(0, np.int64), 'vert_axis': np.zeros(0, np.int64),
                         'vert_df': np.zeros(0, np.float32), 'h': h})
        return np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int32)
    sel = np.concatenate([a for a, _, _ in tri_edge_keys])
    tt = np.concatenate([b for _, b, _ in tri_edge_keys])
    keys = np.concatenate([c for _, _, c in tri_edge_keys])
    order = np.lexsort((tt, sel))                                     # cell-major, table order
    keys = keys[order]
    tri_cell = cells[sel[order]]
    ek = np.unique(keys)
    faces = np.searchsorted(ek, keys).astype(np.int32)
    v0 = (ek // 3).astype(np.int64)
    ax = (ek % 3).astype(np.int64)
    g1 = g[v0].astype(np.int64)
    g1[np.arange(len(ek)), ax] += 1
    v1 = np.searchsorted(vk, lattice_key(g1))
    f0, f1 = f[v0], f[v1]
    t = (f0 / (f0 - f1)).astype(np.float32)
    verts = pos[v0].copy()
    verts[np.arange(len(ek)), ax] = (pos[v0, ax] + t * np.float32(h)).astype(np.float32)
    vert_vkey, vert_axis, vert_df = vk[v0], ax, np.abs(f0 - f1)
    if mask_fn is not None and len(verts):
        keep_v = mask_fn(verts)
        keep_f = keep_v[faces].all(1)
        faces = faces[keep_f]
        used = np.zeros(len(verts), bool)
        used[faces.reshape(-1)] = True
        remap = np.cumsum(used) - 1
        verts = verts[used]
        faces = remap[faces].astype(np.int32)
        tri_cell = tri_cell[keep_f]
        vert_vkey, vert_axis, vert_df = vert_vkey[used], vert_axis[used], vert_df[used]
    if info is not None:
        info.update({'tri_cell': tri_cell, 'vert_vkey': vert_vkey, 'vert_axis': vert_axis, 'vert_df': vert_df, 'h': h})
    return verts.astype(np.float32), faces
