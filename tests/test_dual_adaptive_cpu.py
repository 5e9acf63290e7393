"""oracle/dual_adaptive.py (marching cubes on the adaptive dual graph: cells as large as the level that carries them) -- pinned on
(i) the uniform case, where it must BE oracle/meshing.py bit for bit, and (ii) invariants on mixed-level octrees: closed oriented
meshes of Euler characteristic 2 around a sphere whatever the level pattern, vertices on the level set to the interpolation error
of the local cell size, fewer triangles where the cells are larger."""
import numpy as np
import pytest

from oracle import dual_adaptive as da
from oracle import meshing, spec


def _level(ijk):
    class G:
        pass
    ijk = np.asarray(ijk, np.int32)
    g = G()
    g.ijk, g.n, g.level = ijk, len(ijk), 0
    lo = ijk.min(0) - 1
    shape = ijk.max(0) - lo + 2
    lut = -np.ones(shape, np.int32)
    q = ijk - lo
    lut[q[:, 0], q[:, 1], q[:, 2]] = np.arange(len(ijk))
    g.nbr = np.stack([lut[q[:, 0] + o[0], q[:, 1] + o[1], q[:, 2] + o[2]] for o in spec.NBR_OFFSETS], 1)
    return g


def _block(lo, hi):
    r = np.arange(lo, hi)
    return np.stack(np.meshgrid(r, r, r, indexing='ij'), -1).reshape(-1, 3)


def _edges(faces):
    return np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]])


def _closed_and_oriented(faces):
    e = _edges(faces).astype(np.int64)
    n = int(e.max()) + 1
    fwd = np.unique(e[:, 0] * n + e[:, 1], return_counts=True)
    assert fwd[1].max() == 1                                           # no directed edge twice: consistently oriented, manifold
    assert np.array_equal(fwd[0], np.unique(e[:, 1] * n + e[:, 0]))    # every edge has its opposite: closed


def _euler(verts, faces):
    e = np.unique(np.sort(_edges(faces), 1), axis=0)
    return len(verts) - len(e) + len(faces)


def test_uniform_case_is_the_lattice_mesher_bit_for_bit():
    rs = np.random.RandomState(0)
    ijk = _block(-3, 4)
    ijk = ijk[rs.rand(len(ijk)) < 0.93]                                 # holes: dual cells need all eight voxels
    val = {tuple(c): v for c, v in zip(ijk, rs.randn(len(ijk)).astype(np.float32))}
    w0 = 0.1

    def ev(p):
        q = np.rint(p / np.float32(w0) - 0.5).astype(int)
        return np.array([val[tuple(c)] for c in q], np.float32)
    v0, f0 = meshing.extract(w0, _level(ijk), ev)
    v1, f1 = da.extract(w0, [ijk], ev)
    assert len(f0) > 300
    assert np.array_equal(f0, f1) and np.array_equal(v0, v1)


def _sphere(radius, centre):
    c = np.asarray(centre, np.float32)
    return lambda p: (np.float32(radius) - np.linalg.norm(p - c[None], axis=1)).astype(np.float32)


def _two_level_octree(pattern, seed=0):
    """level 1: an 8^3 block of voxels; level 0: the children of the level-1 voxels `pattern` selects (all eight, or a random
    subset of them -- the missing ones become virtual leaves)"""
    rs = np.random.RandomState(seed)
    l1 = _block(-4, 4)
    if pattern == 'half':
        refine = l1[:, 0] < 0
    elif pattern == 'random':
        refine = rs.rand(len(l1)) < 0.4
    elif pattern == 'none':
        refine = np.zeros(len(l1), bool)
    else:
        refine = np.ones(len(l1), bool)
    ch = (l1[refine][:, None, :] * 2 + da.CORNERS[None]).reshape(-1, 3)
    if pattern == 'random':
        ch = ch[rs.rand(len(ch)) < 0.8]                                 # partial octants
    return [ch, l1]


@pytest.mark.parametrize('pattern', ['half', 'random', 'none', 'all'])
@pytest.mark.parametrize('mise_iter,upsample', [(0, 1), (1, 1), (2, 1), (0, 2), (1, 3)])
def test_sphere_on_mixed_levels_is_closed_with_euler_characteristic_two(pattern, mise_iter, upsample):
    w0 = 0.1
    levels = _two_level_octree(pattern)
    f = _sphere(0.47, (0.013, -0.021, 0.017))                           # inside the block [-0.8, 0.8]^3, off every lattice plane
    info = {}
    v, t = da.extract(w0, levels, f, mise_iter=mise_iter, grid_upsample=upsample, info=info)
    assert len(t) > 100
    _closed_and_oriented(t)
    assert _euler(v, t) == 2
    # outward orientation (f > 0 inside): signed volume positive
    vol = np.einsum('ij,ij->i', v[t[:, 0]], np.cross(v[t[:, 1]], v[t[:, 2]])).sum() / 6
    assert 0.8 * 4 / 3 * np.pi * 0.47 ** 3 < vol < 1.02 * 4 / 3 * np.pi * 0.47 ** 3
    # vertices on the level set to the linear-interpolation error of the coarsest cell in play
    size = 2 * w0 / upsample / (1 << mise_iter)
    assert np.abs(f(v)).max() < 0.6 * size ** 2 / 0.47 + 1e-6


def test_cells_are_as_large_as_their_level():
    w0 = 0.1
    f = _sphere(0.47, (0.013, -0.021, 0.017))
    n = {}
    for pattern in ('all', 'half', 'none'):
        v, t = da.extract(w0, _two_level_octree(pattern), f)
        n[pattern] = len(t)
        _closed_and_oriented(t)
    # a level-1 cell is twice the size: a quarter of the triangles where nothing is refined, in between for the half pattern
    assert 3.3 < n['all'] / n['none'] < 4.7
    assert n['none'] < n['half'] < n['all'] and abs(n['half'] - (n['all'] + n['none']) / 2) < 0.15 * n['all']
    # 'all' = the uniform level-0 mesh of the same region: identical to the lattice mesher on the children
    ch = _two_level_octree('all')[0]
    v0, f0 = meshing.extract(w0, _level(ch), f)
    v1, f1 = da.extract(w0, _two_level_octree('all'), f)
    assert np.array_equal(f0, f1) and np.array_equal(v0, v1)


def test_three_levels_and_virtual_children():
    """levels 0..2, refinement following the surface (what a structure head produces): closed, chi = 2, and the leaf table holds
    the virtual children of partially refined voxels"""
    w0 = 0.05
    f = _sphere(0.31, (0.004, 0.009, -0.006))
    l2 = _block(-3, 3)                                                   # size 0.2 -> [-0.6, 0.6]^3
    def near(c, size, band):
        ctr = (c + 0.5) * size
        return np.abs(f(ctr.astype(np.float32))) < band
    l1 = (l2[near(l2, 0.2, 0.25)][:, None, :] * 2 + da.CORNERS[None]).reshape(-1, 3)
    l1 = l1[near(l1, 0.1, 0.2)]                                          # partial octants at level 1
    l0 = (l1[near(l1, 0.1, 0.09)][:, None, :] * 2 + da.CORNERS[None]).reshape(-1, 3)
    l0 = l0[near(l0, 0.05, 0.06)]
    lv = da.leaves([l0, l1, l2])
    assert len(lv[0]) > len(np.unique(l0, axis=0)) and len(lv[1]) > 0 and len(lv[2]) > 0
    # leaves tile what they cover: no fine voxel in two leaves
    tab = da.Table(da.primal_cells(lv, 1, 0))
    fine = np.concatenate([(c[:, None, :] * (1 << d) + _block(0, 1 << d)[None]).reshape(-1, 3) for d, c in enumerate(lv)])
    assert len(np.unique(fine, axis=0)) == len(fine)
    for m in (0, 1):
        v, t = da.extract(w0, [l0, l1, l2], f, mise_iter=m)
        _closed_and_oriented(t)
        assert _euler(v, t) == 2


def test_on_a_solved_field_with_two_adaptive_levels_the_mesh_is_a_closed_sphere():
    """A real hierarchy and a real field (oracle pipeline, adaptive_depth 2, samples sparser than the finest voxels -- the case of
    tests/test_gpu_parity.py::test_adaptive_depth_meshing_covers_what_the_finest_level_leaves_open): the level-0 voxels alone
    leave holes, the adaptive dual graph over levels 0 and 1 (childless level-1 voxels + the virtual children of the refined ones)
    closes them, with as many triangles as the lattice mesher where everything ends up at level-0 resolution."""
    from oracle import pipeline
    n = 700
    k = np.arange(n) + 0.5
    phi, z = np.pi * (1 + 5 ** 0.5) * k, 1 - 2 * k / n
    nrm = np.stack([np.cos(phi) * np.sqrt(1 - z * z), np.sin(phi) * np.sqrt(1 - z * z), z], 1).astype(np.float32)
    xyz = (nrm * np.float32(0.45 * 2.5)).astype(np.float32)
    fld = pipeline.reconstruct(xyz, nrm, adaptive_depth=2, tol=1e-6)
    ev = lambda p: pipeline.evaluate(fld, p)[0]
    lv = [L.ijk for L in fld['hier'].levels[:2]]
    leaf = da.leaves(lv)
    assert len(leaf[0]) > len(lv[0]) and 0 < len(leaf[1]) < len(lv[1])          # virtual children; some level-1 voxels are leaves
    v0, t0 = da.extract(fld['voxel_size'], lv[:1], ev)
    e = np.sort(_edges(t0), 1)
    assert (np.unique(e, axis=0, return_counts=True)[1] == 1).sum() > 0          # level 0 alone: holes
    for m in (0, 1):
        v, t = da.extract(fld['voxel_size'], lv, ev, mise_iter=m)
        _closed_and_oriented(t)
        assert _euler(v, t) == 2
        if m == 0:
            assert len(t) == len(pipeline.extract_dual_mesh(fld, mise_iter=0)[1])
            assert np.array_equal(t, pipeline.extract_dual_mesh(fld, mise_iter=0, dual_graph='adaptive')[1])
