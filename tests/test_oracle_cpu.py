"""CPU tests: the oracle against (i) an independent solver, (ii) reference-independent
invariants and (iii) the committed golden fixtures (SURVEY.md section 8c substitute pins).
No GPU needed.  The reference holds no golden vectors for this path -- parity is unpinned with
respect to the (absent) reference implementation and these tests say so by construction."""
import os

import numpy as np
import pytest

from conftest import make_cloud

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def test_morton_roundtrip_and_level_shift():
    from oracle import spec
    rs = np.random.RandomState(0)
    ijk = rs.randint(-2 ** 19, 2 ** 19, size=(5000, 3)).astype(np.int32)
    k0 = spec.morton_key(ijk, 0)
    assert np.array_equal(spec.morton_decode(k0, 0), ijk)
    for d in (1, 2, 3, 4):
        assert np.array_equal(spec.morton_key(ijk >> d, d), k0 >> (3 * d))
    # maximum coordinates are representable, one past is rejected
    spec.morton_key(np.array([[2 ** 20 - 1, -2 ** 20, 0]], np.int32), 0)
    with pytest.raises(AssertionError):
        spec.morton_key(np.array([[2 ** 20, 0, 0]], np.int32), 0)


def test_bspline_partition_of_unity():
    from oracle import spec
    u = np.linspace(0, 1, 101).astype(np.float32)
    w, dw = spec.bspline3(u)
    np.testing.assert_allclose(w.sum(1), 1.0, atol=1e-6)
    np.testing.assert_allclose(dw.sum(1), 0.0, atol=1e-6)
    assert (w >= 0).all()


def test_hierarchy_parent_property_and_nbr_symmetry():
    from oracle import hierarchy
    xyz, _ = make_cloud('torus', 2000, 0.005, 1)
    for builder in ('build_point_splatting', 'build_point_neighborhood'):
        h = getattr(hierarchy.Hierarchy(0.05, 4), builder)(xyz)
        for d in range(3):
            assert (h.levels[d].parent >= 0).all(), 'every voxel has an active parent'
        L = h.levels[0]
        for s in range(27):
            j = L.nbr[:, s]
            ok = j >= 0
            assert np.array_equal(L.nbr[j[ok], 26 - s], np.nonzero(ok)[0])


def test_voxel_status_classes_of_an_adaptive_hierarchy():
    """oracle.hierarchy.evaluate_voxel_status (models/loss.py:155): 0 outside, 1 voxel without children, 2 voxel with children --
    checked through the parent links the hierarchy already holds, on a ground-truth structure that stops early in flat regions."""
    from oracle import hierarchy
    xyz, nrm = make_cloud('torus', 4000, 0.005, 5)
    xyz = xyz * np.float32(2.0)
    gt = hierarchy.Hierarchy(0.1, 4).build_adaptive_normal_variation(xyz, nrm, tau=0.05, adaptive_depth=2)
    cand = hierarchy.Hierarchy(0.1, 4).build_point_neighborhood(xyz)
    classes = set()
    for d in range(4):
        q = cand.levels[d].ijk
        st = gt.evaluate_voxel_status(q, d)
        inside = gt.levels[d].lookup(q) >= 0
        assert np.array_equal(st > 0, inside) and inside.sum() == gt.levels[d].n
        if d == 0:
            assert not (st == 2).any()                    # the finest level has no children
        else:
            kids = np.zeros(gt.levels[d].n, bool)
            kids[gt.levels[d - 1].parent] = True          # parent index of every finer voxel
            assert np.array_equal(st[inside] == 2, kids[gt.levels[d].lookup(q[inside])])
        classes |= set(np.unique(st).tolist())
    assert classes == {0, 1, 2}
    # own grid: never class 0; a foreign level far away: all 0
    assert (gt.evaluate_voxel_status(gt.levels[2].ijk, 2) > 0).all()
    assert not gt.evaluate_voxel_status(gt.levels[2].ijk + 1000, 2).any()


def test_mc_table_matches_product_header_and_is_watertight():
    import re
    from oracle import mc_tables as m
    hdr = open(os.path.join(os.path.dirname(GOLD), '..', 'nksr_amd', 'csrc', 'mc_table.h')).read()
    rows = re.findall(r'\{([-\d,]+)\},', hdr)
    T = np.array([[int(x) for x in r.split(',')] for r in rows]).reshape(256, -1, 3)
    assert np.array_equal(T, m.TRI_TABLE)
    assert m.TRI_COUNT.max() == 5 and m.TRI_COUNT[0] == 0 and m.TRI_COUNT[255] == 0
    # every shared face of two random neighbouring cells is cut identically: count the segments
    rs = np.random.RandomState(0)
    f = rs.randn(7, 7, 7).astype(np.float32)
    from oracle import meshing, spec

    class G:  # minimal stand-in for a level with a dense 7^3 block of voxels
        pass
    ijk = np.stack(np.meshgrid(*[np.arange(7)] * 3, indexing='ij'), -1).reshape(-1, 3).astype(np.int32)
    g = G()
    g.ijk, g.n = ijk, len(ijk)
    lut = -np.ones((9, 9, 9), np.int32)
    lut[ijk[:, 0] + 1, ijk[:, 1] + 1, ijk[:, 2] + 1] = np.arange(len(ijk))
    g.nbr = np.stack([lut[ijk[:, 0] + 1 + o[0], ijk[:, 1] + 1 + o[1], ijk[:, 2] + 1 + o[2]] for o in spec.NBR_OFFSETS], 1)

    def ev(p):
        q = np.rint(p / 1.0 - 0.5).astype(int)
        return f[q[:, 0], q[:, 1], q[:, 2]]
    v, t = meshing.extract(1.0, g, ev, mise_iter=0)
    e = np.sort(np.concatenate([t[:, [0, 1]], t[:, [1, 2]], t[:, [2, 0]]]), 1)
    ue, cnt = np.unique(e, axis=0, return_counts=True)
    assert cnt.max() == 2
    # boundary edges only on the outer faces of the block
    b = np.unique(ue[cnt == 1])
    on_boundary = ((v[b] <= 0.5 + 1e-6) | (v[b] >= 6.5 - 1e-6)).any(1)
    assert on_boundary.all()


@pytest.fixture(scope='module')
def sphere_field():
    from oracle import pipeline
    xyz, nrm = make_cloud('sphere', 3000, 0.005, 0)
    xs = (xyz * np.float32(2.0)).astype(np.float32)
    return xyz, nrm, xs, pipeline.reconstruct(xs, nrm, tol=1e-6)


def test_system_is_spd_and_pcg_matches_scipy(sphere_field):
    import scipy.sparse.linalg as sla
    from oracle import solve
    xyz, nrm, xs, fld = sphere_field
    A, b = fld['A'], fld['b']
    assert abs(A - A.T).max() <= 1e-6 * abs(A).max()
    assert (A.diagonal() > 0).all()
    rs = np.random.RandomState(0)
    for _ in range(5):
        x = rs.randn(A.shape[0])
        assert x @ (A.astype(np.float64) @ x) > 0
    assert fld['rel'] <= 1e-6
    xs_, info = sla.cg(A.astype(np.float64), b.astype(np.float64), rtol=1e-12, maxiter=20000)
    assert info == 0
    assert abs(fld['alpha'] - xs_).max() <= 1e-3 * abs(xs_).max()
    # one SpMV against scipy's CSR product
    y = solve.csr_spmv(A.indptr, A.indices, A.data, fld['alpha'])
    np.testing.assert_allclose(y, A @ fld['alpha'], rtol=1e-4, atol=1e-5 * abs(y).max())


def test_field_invariants_on_sphere(sphere_field):
    from oracle import pipeline
    xyz, nrm, xs, fld = sphere_field
    f, g = pipeline.evaluate(fld, xs, grad=True)
    gn = np.linalg.norm(g, axis=1)
    assert np.abs(f).mean() < 0.02 * gn.mean() * 0.1 * 5          # |f| small at the inputs (in voxel units)
    cosang = (-g * nrm).sum(1) / gn
    assert cosang.mean() > 0.98                                     # -grad f aligned with the outward normals
    fi, _ = pipeline.evaluate(fld, xs - (0.08 * nrm).astype(np.float32))
    fo, _ = pipeline.evaluate(fld, xs + (0.08 * nrm).astype(np.float32))
    assert (fi > 0).mean() > 0.99 and (fo < 0).mean() > 0.99       # f > 0 inside


def test_mesh_is_closed_genus0_and_accurate(sphere_field):
    from oracle import pipeline
    xyz, nrm, xs, fld = sphere_field
    v, t = pipeline.extract_dual_mesh(fld, mise_iter=0)
    e = np.sort(np.concatenate([t[:, [0, 1]], t[:, [1, 2]], t[:, [2, 0]]]), 1)
    ue, cnt = np.unique(e, axis=0, return_counts=True)
    assert (cnt == 2).all()
    assert len(v) - len(ue) + len(t) == 2
    r = np.linalg.norm(v / 2.0, axis=1)
    assert abs(np.median(r) - 0.45) < 0.005 and r.std() < 0.01
    # outward orientation: triangle normals point away from the centre
    n = np.cross(v[t[:, 1]] - v[t[:, 0]], v[t[:, 2]] - v[t[:, 0]])
    assert ((n * v[t].mean(1)).sum(1) > 0).mean() > 0.999
    # MISE keeps the mesh closed: hanging vertices take the coarse interpolant (no T-junction cracks)
    for mi, grow in ((1, 3), (2, 12)):
        v1, t1 = pipeline.extract_dual_mesh(fld, mise_iter=mi)
        assert len(t1) > grow * len(t)
        e1 = np.sort(np.concatenate([t1[:, [0, 1]], t1[:, [1, 2]], t1[:, [2, 0]]]), 1)
        ue1, cnt1 = np.unique(e1, axis=0, return_counts=True)
        assert (cnt1 == 2).all() and len(v1) - len(ue1) + len(t1) == 2


@pytest.mark.parametrize('name', ['bunny_2k', 'sphere_3k'])
def test_oracle_reproduces_golden(name):
    from oracle import make_golden
    g = np.load(os.path.join(GOLD, name + '_golden.npz'))
    if name == 'bunny_2k':
        d = np.load(os.path.join(GOLD, 'bunny_2k.npz'))
        xyz, nrm = d['xyz'], d['normal']
    else:
        from nksr_amd import utils
        xyz, nrm = utils.synth_sphere(3000, 0.45, 0.005, seed=0)
    out = make_golden.run_case(xyz, nrm, float(g['voxel_size']))
    for d in range(4):
        assert np.array_equal(out['keys_%d' % d], g['keys_%d' % d])
    assert int(out['A_nnz']) == int(g['A_nnz'])
    np.testing.assert_allclose(out['A_diag'], g['A_diag'], rtol=1e-5)
    np.testing.assert_allclose(out['alpha'], g['alpha'], rtol=0, atol=1e-4 * abs(g['alpha']).max())
    for mise in (0, 1):
        assert np.array_equal(out['mesh%d_f' % mise], g['mesh%d_f' % mise])
        assert np.array_equal(out['mesh%d_vert_vkey' % mise], g['mesh%d_vert_vkey' % mise])
        np.testing.assert_allclose(out['mesh%d_v' % mise], g['mesh%d_v' % mise], atol=1e-5)
