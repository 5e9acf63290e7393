"""ORACLE (test infrastructure only -- never imported by the product).
Dual marching cubes on the ADAPTIVE dual graph: cells as large as the hierarchy level that carries them.

What it restates: field.extract_dual_mesh(mise_iter, grid_upsample) of the reference (call sites examples/recons_simple.py:27,
recons_scannet.py:29, models/nksr_net.py:214,284) for a hierarchy whose structure head stops early somewhere
(LayerField(dec_svh, adaptive_depth), models/nksr_net.py:132): the reference flattens the levels below adaptive_depth into their
LEAVES (voxels without children), builds the dual graph of that octree -- one hexahedron per octree corner, its eight corners the
centres of the leaves around it -- and runs marching cubes on it ("build dual grid of svh", V ~ 2-4 x
surface voxels).  The implementation is in the absent wheel; the algorithm restated here is the published one (Schaefer & Warren,
Dual Marching Cubes: Primal Contouring of Dual Grids, 2004) -- PARITY UNPINNED like the rest of row a9, pinned instead on
(i) the uniform case: with one level, grid_upsample = 1 and mise_iter = 0 it IS oracle/meshing.py, bit for bit, vertex order and
triangle order included, and (ii) invariants on mixed-level hierarchies: closed orientable meshes with Euler characteristic 2 on a
sphere whatever the level pattern, vertices on the level set to the interpolation error of the local cell size.

Specification ("adaptive dual graph"):
  * octree: the voxels of levels < adaptive_depth; a voxel with at least one child is internal and ALL eight of its children are
    nodes (a child the hierarchy does not hold is a "virtual" leaf: empty space next to a refined region still carries a sample);
    leaves = level-0 voxels, childless voxels, virtual children;
  * primal cells: a leaf of level d is split into U^3 cells (grid_upsample); a cell is (lam, C): size 2^lam fine units, minimum
    corner C * 2^lam, fine unit u = w0 / (U 2^M), a level-d leaf starts at lam = d + M (M = mise_iter);
  * dual cell: for every corner k of any cell, the eight cells containing the fine voxels k - 1 + o, o in {0,1}^3 (corner order
    c = 4 ox + 2 oy + oz as in oracle/meshing.py); it exists when all eight do; a cell may fill several corners (k on a face or an
    edge of a larger cell: a degenerate hexahedron).  Dual cells are ordered by the Morton key of k - 1;
  * samples: f at the cell centres, x = fl(fl(C 2^lam * u) + 2^lam * u / 2) (two fp32 roundings, the lattice positions of
    oracle/meshing.py in the uniform case);
  * MISE: M times, every cell that is a corner of a sign-changing dual cell is split into 8 and the dual graph is rebuilt (no
    hanging-vertex rule: the dual of an octree is conforming by construction);
  * 256-case table of oracle/mc_tables.py per dual cell; a mesh vertex is named by the ORDERED pair of cells (A, B) its cube edge
    joins (A on the low side of the edge's axis) and lies at pA + t (pB - pA), t = fA / (fA - fB), pB - pA taken exactly from the
    integer cell coordinates; triangles that name a vertex twice (collapsed edges of degenerate hexahedra) are dropped, vertices no
    triangle uses are dropped; vertices ordered by (A, axis, B) -- key A 2^33 + axis 2^31 + B --, triangles by dual cell, then
    table order.
"""
import numpy as np

from . import mc_tables, meshing, spec

CORNERS = spec.CORNER_OFFSETS.astype(np.int64)


def _sorted_unique(coords):
    """coords [n,3] -> (sorted unique keys, coords in that order)"""
    k = np.unique(meshing.lattice_key(coords))
    return k, meshing.lattice_decode(k).astype(np.int64)


def leaves(level_ijk):
    """level_ijk[d]: [n_d,3] integer coordinates of the level-d voxels, d < adaptive_depth.  -> list over d of [m_d,3] leaf coordinates
    (hierarchy leaves and virtual children), sorted by key."""
    D = len(level_ijk)
    keys = [np.unique(meshing.lattice_key(np.asarray(c, np.int64).reshape(-1, 3))) for c in level_ijk]
    out = [[] for _ in range(D)]
    for d in range(D):
        c = meshing.lattice_decode(keys[d]).astype(np.int64)
        if d == 0:
            out[0].append(c)
            continue
        ch = (c[:, None, :] * 2 + CORNERS[None]).reshape(-1, 3)
        have = (meshing._find(keys[d - 1], meshing.lattice_key(ch)) >= 0).reshape(-1, 8)
        internal = have.any(1)
        out[d].append(c[~internal])
        out[d - 1].append(ch.reshape(-1, 8, 3)[internal][~have[internal]])      # virtual children of refined voxels
    return [_sorted_unique(np.concatenate(o))[1] if sum(len(x) for x in o) else np.zeros((0, 3), np.int64) for o in out]


def primal_cells(leaf_coords, upsample, mise_iter):
    """-> {lam: [n,3] cell coordinates}"""
    U = int(upsample)
    sub = np.array([[a, b, c] for a in range(U) for b in range(U) for c in range(U)], np.int64)
    cells = {}
    for d, c in enumerate(leaf_coords):
        if len(c):
            cells[d + mise_iter] = (c[:, None, :] * U + sub[None]).reshape(-1, 3)
    return cells


class Table:
    """The cells of all sizes as one table, smallest cells first, each size sorted by key: id = offset[lam] + rank."""

    def __init__(self, cells):
        self.lams = sorted(l for l in cells if len(cells[l]))
        self.keys, self.coords, self.offset = {}, {}, {}
        n = 0
        for l in self.lams:
            self.keys[l], self.coords[l] = _sorted_unique(cells[l])
            self.offset[l] = n
            n += len(self.keys[l])
        self.n = n
        self.lam = np.concatenate([np.full(len(self.keys[l]), l, np.int64) for l in self.lams]) if n else np.zeros(0, np.int64)
        self.C = np.concatenate([self.coords[l] for l in self.lams]) if n else np.zeros((0, 3), np.int64)

    def find(self, fine):
        """id of the cell containing each fine voxel (-1: none); the smallest size is asked first"""
        out = -np.ones(len(fine), np.int64)
        for l in self.lams:
            todo = np.nonzero(out < 0)[0]
            if not len(todo):
                break
            j = meshing._find(self.keys[l], meshing.lattice_key(fine[todo] >> l))
            out[todo[j >= 0]] = self.offset[l] + j[j >= 0]
        return out

    def positions(self, u):
        s = (np.int64(1) << self.lam)
        g = (self.C * s[:, None]).astype(np.float32)
        half = s.astype(np.float32) * (np.float32(0.5) * np.float32(u))
        return ((g * np.float32(u)).astype(np.float32) + half[:, None]).astype(np.float32)


def dual_cells(tab):
    """-> (corner coordinates k - 1 [n,3] of the dual cells in their order, cidx [n,8])"""
    if tab.n == 0:
        return np.zeros((0, 3), np.int64), np.zeros((0, 8), np.int64)
    s = (np.int64(1) << tab.lam)
    k = ((tab.C[:, None, :] + CORNERS[None]) * s[:, None, None]).reshape(-1, 3) - 1
    _, km1 = _sorted_unique(k)
    cidx = np.stack([tab.find(km1 + CORNERS[c][None]) for c in range(8)], 1)
    ok = (cidx >= 0).all(1)
    return km1[ok], cidx[ok]


def extract(voxel_size, level_ijk, eval_fn, mise_iter=0, grid_upsample=1, mask_fn=None, info=None):
    """eval_fn(xyz [n,3] f32) -> f [n] f32; mask_fn(xyz) -> bool [n] (True = keep).  -> (verts [V,3] f32, faces [T,3] int32)."""
    M, U = int(mise_iter), int(grid_upsample)
    u = np.float32(float(voxel_size) / U / (1 << M))
    cells = primal_cells(leaves(level_ijk), U, M)
    empty = np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int32)
    for m in range(M + 1):
        tab = Table(cells)
        pos = tab.positions(u)
        f = eval_fn(pos).astype(np.float32) if tab.n else np.zeros(0, np.float32)
        km1, cidx = dual_cells(tab)
        inside = f > 0
        config = (inside[cidx] * (1 << np.arange(8))[None]).sum(1).astype(np.int32) if len(cidx) else np.zeros(0, np.int32)
        if info is not None:
            info.setdefault('levels', []).append({'lam': tab.lam.copy(), 'C': tab.C.copy(), 'pos': pos, 'f': f, 'km1': km1, 'cidx': cidx,
                                                  'config': config})
        if m < M:
            act = (config != 0) & (config != 255)
            split = np.zeros(tab.n, bool)
            split[cidx[act].reshape(-1)] = True
            split &= tab.lam > 0
            cells = {}
            for l in tab.lams:
                sel = slice(tab.offset[l], tab.offset[l] + len(tab.keys[l]))
                keep, sp = tab.coords[l][~split[sel]], tab.coords[l][split[sel]]
                if len(keep):
                    cells.setdefault(l, []).append(keep)
                if len(sp):
                    cells.setdefault(l - 1, []).append((sp[:, None, :] * 2 + CORNERS[None]).reshape(-1, 3))
            cells = {l: np.concatenate(v) for l, v in cells.items()}
    if not len(cidx):
        return empty
    ntri = mc_tables.TRI_COUNT[config]
    tris = []
    for t in range(mc_tables.TRI_TABLE.shape[1]):
        sel = np.nonzero(ntri > t)[0]
        e = mc_tables.TRI_TABLE[config[sel], t]                       # [k,3] cube edges
        lo, ax = mc_tables.EDGE_LO[e], mc_tables.EDGE_AXIS[e].astype(np.int64)
        hi = lo | (4 >> ax)
        a, b = cidx[sel[:, None], lo], cidx[sel[:, None], hi]
        tris.append((sel, np.full(len(sel), t), (a << 33) | (ax << 31) | b))
    sel = np.concatenate([x for x, _, _ in tris])
    if not len(sel):
        return empty
    tt = np.concatenate([x for _, x, _ in tris])
    keys = np.concatenate([x for _, _, x in tris])
    order = np.lexsort((tt, sel))
    keys, tri_cell = keys[order], sel[order]
    good = (keys[:, 0] != keys[:, 1]) & (keys[:, 1] != keys[:, 2]) & (keys[:, 0] != keys[:, 2])
    keys, tri_cell = keys[good], tri_cell[good]
    ek = np.unique(keys)
    faces = np.searchsorted(ek, keys).astype(np.int32)
    A, B = ek >> 33, ek & 0x7FFFFFFF
    fa, fb = f[A], f[B]
    t = (fa / (fa - fb)).astype(np.float32)
    sA, sB = np.int64(1) << tab.lam[A], np.int64(1) << tab.lam[B]
    d2 = 2 * (tab.C[B] * sB[:, None] - tab.C[A] * sA[:, None]) + (sB - sA)[:, None]          # doubled centre difference, fine units
    d = (d2.astype(np.float32) * (np.float32(0.5) * u)).astype(np.float32)
    verts = (pos[A] + (t[:, None] * d).astype(np.float32)).astype(np.float32)
    if mask_fn is not None and len(verts):
        keep_f = mask_fn(verts)[faces].all(1)
        faces, tri_cell = faces[keep_f], tri_cell[keep_f]
        used = np.zeros(len(verts), bool)
        used[faces.reshape(-1)] = True
        remap = np.cumsum(used) - 1
        verts, ek = verts[used], ek[used]
        faces = remap[faces].astype(np.int32)
    if info is not None:
        info.update({'tri_cell': tri_cell, 'vert_pair': ek, 'u': u})
    return verts, faces
