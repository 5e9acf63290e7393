"""GPU parity: every HIP stage against the CPU oracle on the same seeded inputs.
Bar: index-exact for voxel sets / neighbour tables / mesh topology,
fp32 tolerances (stated per test) for kernel rows, the matrix, PCG and field values."""
import numpy as np
import pytest
import torch

from conftest import make_cloud
import parity_util as pu

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device('cuda:0')


def _setup(kind='sphere', n=3000, vs=0.05, K=4, H=16, depth=4, init_scale=0.5, seed=0, random_feats=True):
    """Same hierarchy / features / interpolators on both sides."""
    import nksr_amd
    from nksr_amd import configs
    from nksr_amd.nn.network import NKSRNetwork
    from oracle import hierarchy, kernel
    xyz, nrm = make_cloud(kind, n, 0.005, seed)
    scale = 0.1 / vs
    xyz = (xyz * np.float32(scale)).astype(np.float32)
    hp = configs.get_hparams('ks', kernel_dim=K, interpolator={'n_hidden': 2, 'hidden_dim': H}, tree_depth=depth,
                             interpolator_init_scale=init_scale)
    net = NKSRNetwork(hp)
    rs = np.random.RandomState(seed + 1)
    for it in net.interpolators:  # non-trivial biases so every MLP branch is exercised
        it.b1.data = torch.from_numpy(rs.randn(H).astype(np.float32) * 0.2)
        it.b2.data = torch.from_numpy(rs.randn(H).astype(np.float32) * 0.2)
        it.b3.data = torch.from_numpy(rs.randn(K).astype(np.float32) * 0.05)
    oh = hierarchy.Hierarchy(0.1, depth).build_point_neighborhood(xyz)
    svh = nksr_amd.SparseFeatureHierarchy(0.1, depth, _dev()).build_point_neighborhood(torch.from_numpy(xyz).to(_dev()))
    feats = []
    for L in oh.levels:
        f = np.zeros((L.n, K), np.float32)
        f[:, 0] = 1
        if random_feats:
            f += rs.randn(L.n, K).astype(np.float32) * 0.3
        feats.append(f)
    ointerps = [kernel.Interpolator(*[p.detach().numpy() for p in (i.W1, i.b1, i.W2, i.b2, i.W3, i.b3)]) for i in net.interpolators]
    return xyz, nrm, oh, svh, feats, ointerps, net


def test_hierarchy_exact():
    import nksr_amd
    from oracle import hierarchy
    xyz, nrm = make_cloud('torus', 5000, 0.005, 3)
    xyz = xyz * np.float32(2.0)
    for builder in ('build_point_splatting', 'build_point_neighborhood'):
        oh = getattr(hierarchy.Hierarchy(0.1, 4), builder)(xyz)
        svh = getattr(nksr_amd.SparseFeatureHierarchy(0.1, 4, _dev()), builder)(torch.from_numpy(xyz).to(_dev()))
        for d in range(4):
            g = svh.level(d)
            assert g.num_voxels == oh.levels[d].n
            assert np.array_equal(g.keys.cpu().numpy(), oh.levels[d].keys)
            assert np.array_equal(g.ijk.cpu().numpy(), oh.levels[d].ijk)
            assert np.array_equal(g.nbr.cpu().numpy(), oh.levels[d].nbr)
            np.testing.assert_array_equal(svh.get_voxel_centers(d).cpu().numpy(), oh.levels[d].centers())


def test_voxel_status_and_visualization_match_the_oracle():
    """SparseFeatureHierarchy.evaluate_voxel_status (models/loss.py:155: structure ground truth) on a query grid that is larger
    than the hierarchy's own (the decoder's candidate grid), and get_visualization (models/nksr_net.py:71)."""
    import nksr_amd
    from oracle import hierarchy
    xyz, nrm = make_cloud('torus', 4000, 0.005, 5)
    xyz = xyz * np.float32(2.0)
    dev = _dev()
    tx, tn = torch.from_numpy(xyz).to(dev), torch.from_numpy(nrm).to(dev)
    gt = nksr_amd.SparseFeatureHierarchy(0.1, 4, dev).build_adaptive_normal_variation(tx, tn, tau=0.05, adaptive_depth=2)
    ogt = hierarchy.Hierarchy(0.1, 4).build_adaptive_normal_variation(xyz, nrm, tau=0.05, adaptive_depth=2)
    cand = nksr_amd.SparseFeatureHierarchy(0.1, 4, dev).build_point_neighborhood(tx)
    seen = set()
    for d in range(4):
        assert np.array_equal(gt.level(d).keys.cpu().numpy(), ogt.levels[d].keys)
        g = cand.grids[d]
        got = gt.evaluate_voxel_status(g, d).cpu().numpy()
        want = ogt.evaluate_voxel_status(g.ijk.cpu().numpy(), d)
        assert np.array_equal(got, want)
        seen |= set(np.unique(got).tolist())
        assert (got > 0).sum() == gt.num_voxels(d)          # the candidate grid covers every ground-truth voxel
    assert seen == {0, 1, 2}                                 # adaptive structure: stopped voxels exist next to refined ones
    vis = gt.get_visualization()
    assert len(vis) == 4
    for d, (centres, size) in enumerate(vis):
        assert size == pytest.approx(0.1 * (1 << d))
        np.testing.assert_array_equal(centres, ogt.levels[d].centers())


def test_grid_accessors_of_the_reference_surface():
    """``grids[d].active_grid_coords()`` / ``grid_to_world`` / ``voxel_size`` (models/loss.py:36,45-46), ``world_to_grid``,
    ``ijk_to_index``, ``build_from_grid_coords`` and ``build_from_keys`` against the oracle's levels."""
    import nksr_amd
    from oracle import hierarchy
    xyz, _ = make_cloud('sphere', 3000, 0.005, 11)
    dev = _dev()
    oh = hierarchy.Hierarchy(0.1, 3).build_point_splatting(xyz)
    svh = nksr_amd.SparseFeatureHierarchy(0.1, 3, dev).build_point_splatting(torch.from_numpy(xyz).to(dev))
    rebuilt = nksr_amd.SparseFeatureHierarchy(0.1, 3, dev)
    for d in range(3):
        g, lv = svh.grids[d], oh.levels[d]
        ijk = g.active_grid_coords()
        assert np.array_equal(ijk.cpu().numpy(), lv.ijk) and g.voxel_size == pytest.approx(lv.voxel_size)
        world = g.grid_to_world(ijk.float())
        np.testing.assert_array_equal(world.cpu().numpy(), lv.centers())
        np.testing.assert_allclose(g.world_to_grid(world).cpu().numpy(), lv.ijk.astype(np.float32), atol=2e-4)
        # canonical index of coordinates: identity on the active voxels, -1 one lattice step outside the bounding box
        perm = torch.randperm(g.num_voxels, device=dev)
        assert torch.equal(g.ijk_to_index(ijk[perm]).long(), perm)
        outside = ijk.max(0).values[None] + torch.tensor([[1, 0, 0], [0, 2, 0]], dtype=ijk.dtype, device=dev)
        assert g.ijk_to_index(outside).tolist() == [-1, -1]
        rebuilt.build_from_grid_coords(d, ijk[perm])          # any order, duplicates allowed
    for d in range(3):
        assert torch.equal(rebuilt.level(d).keys, svh.level(d).keys) and torch.equal(rebuilt.level(d).nbr, svh.level(d).nbr)
    again = nksr_amd.SparseFeatureHierarchy(0.1, 3, dev).build_from_keys([svh.level(d).keys.flip(0) for d in range(3)])
    assert all(torch.equal(again.level(d).keys, svh.level(d).keys) for d in range(3))


def test_empty_and_tiny_inputs():
    import nksr_amd
    svh = nksr_amd.SparseFeatureHierarchy(0.1, 3, _dev()).build_point_splatting(torch.zeros((0, 3), device=_dev()))
    assert svh.grids == [None, None, None] and svh.num_unknowns == 0
    one = torch.tensor([[0.01, -0.02, 0.03]], device=_dev())
    svh = nksr_amd.SparseFeatureHierarchy(0.1, 3, _dev()).build_point_splatting(one)
    assert [svh.num_voxels(d) for d in range(3)] == [8, 8, 8]
    svh = nksr_amd.SparseFeatureHierarchy(0.1, 3, _dev()).build_point_neighborhood(one)
    assert [svh.num_voxels(d) for d in range(3)] == [27, 27, 27]


@pytest.mark.parametrize('K,H', [(4, 16), (16, 32)])
@pytest.mark.parametrize('approx', [False, True])
def test_kernel_rows(K, H, approx):
    from nksr_amd.fields import KernelField
    from oracle import kernel
    xyz, nrm, oh, svh, feats, ointerps, net = _setup(n=1500, K=K, H=H)
    fld = KernelField(svh, net.interpolators, [torch.from_numpy(f) for f in feats], approx_kernel_grad=approx)
    psis = [kernel.voxel_psi(feats[d], ointerps[d]) for d in range(4)]
    for d in range(4):
        np.testing.assert_allclose(fld._psi[d].cpu().numpy(), psis[d], rtol=2e-5, atol=2e-6)
    q = np.concatenate([xyz[:700], oh.levels[0].centers()[:300]])
    cols, val, dval = kernel.kernel_rows(oh, feats, ointerps, psis, q, True, approx)
    gv, gd = fld.kernel_rows(torch.from_numpy(q).to(_dev()), grad=True)
    np.testing.assert_allclose(gv.cpu().numpy(), val, rtol=1e-4, atol=2e-6)
    # gradients carry a 1/w factor (up to 10 in model units): tolerance scaled accordingly
    np.testing.assert_allclose(gd.cpu().numpy(), dval, rtol=1e-4, atol=5e-5)


def test_assembly_and_spmv():
    import scipy.sparse as sp
    from nksr_amd import solver
    from nksr_amd.fields import KernelField
    from oracle import solve
    xyz, nrm, oh, svh, feats, ointerps, net = _setup(n=3000)
    fld = KernelField(svh, net.interpolators, [torch.from_numpy(f) for f in feats])
    nxyz = oh.levels[0].centers()
    nval = np.random.RandomState(5).randn(len(nxyz), 3).astype(np.float32)
    wp, wn = 1e4 / len(xyz), 1e4 / len(nxyz) * 0.01
    A, b, G, Q, psis = solve.assemble(oh, feats, ointerps, xyz, nxyz, nval, wp, wn, 1.0)
    t = lambda a: torch.from_numpy(a).to(_dev())
    rowptr, cols, vals, diag, gb = fld.assemble(t(xyz), t(nxyz), t(nval), wp, wn, 1.0)
    M = A.shape[0]
    lc, lv = solver.csr_logical(rowptr, cols, vals)
    Ag = sp.csr_matrix((lv.cpu().numpy(), lc.cpu().numpy(), rowptr.cpu().numpy()), shape=(M, M))
    # exactly symmetric, sorted columns, SPD diagonal
    assert abs(Ag - Ag.T).max() == 0.0
    Ag.sort_indices()
    D = (Ag - A).tocoo()
    scale = abs(A).max()
    assert abs(D.data).max() <= 2e-5 * scale, abs(D.data).max() / scale
    # structural zeros dropped on the GPU side only where the oracle value is ~0 as well
    np.testing.assert_allclose(gb.cpu().numpy(), b, rtol=1e-4, atol=1e-5 * abs(b).max())
    np.testing.assert_allclose(diag.cpu().numpy(), A.diagonal(), rtol=1e-4)
    x = np.random.RandomState(1).randn(M).astype(np.float32)
    y = solver.spmv(rowptr, cols, vals, t(x)).cpu().numpy()
    yo = solve.csr_spmv(Ag.indptr, Ag.indices, Ag.data, x)
    # same matrix, fp32: |dy| <= 1e-5 * |A||x|  (rel-tol 1e-5 on SpMV)
    bound = 1e-5 * (abs(Ag) @ abs(x))
    assert (abs(y - yo) <= bound + 1e-30).all()
    # both physical layouts (include/nksr_hip.h col_format): packed 21-bit columns is the default here, the
    # int32 layout (used when M > 2^21) must hold the same matrix and give the same product and solve
    assert solver.col_format(cols) == 1
    fld.solver_config['col_format'] = 0
    rowptr0, cols0, vals0, diag0, gb0 = fld.assemble(t(xyz), t(nxyz), t(nval), wp, wn, 1.0)
    assert solver.col_format(cols0) == 0 and torch.equal(rowptr0, rowptr) and torch.equal(diag0, diag) and torch.equal(gb0, gb)
    lc0, lv0 = solver.csr_logical(rowptr0, cols0, vals0)
    assert torch.equal(lc0, lc) and torch.equal(lv0, lv)
    y0 = solver.spmv(rowptr0, cols0, vals0, t(x)).cpu().numpy()
    assert (abs(y0 - yo) <= bound + 1e-30).all()
    s1 = solver.pcg_solve(rowptr, cols, vals, diag, gb, tol=1e-6)
    s0 = solver.pcg_solve(rowptr0, cols0, vals0, diag0, gb0, tol=1e-6)
    assert s1[2] <= 1e-6 and s0[2] <= 1e-6
    assert float((s1[0] - s0[0]).abs().max()) <= 1e-4 * float(s1[0].abs().max())


@pytest.mark.parametrize('row_format', ['factors', 'dense'])
@pytest.mark.parametrize('approx', [False, True])
def test_fused_operator_matches_the_assembled_matrix(approx, row_format, monkeypatch):
    """fused_mode=True (examples/recons_waymo.py:33): the matrix-free operator, its right-hand side and diagonal against
    the assembled CSR of the same system and against the oracle's matrix; fixed-iteration PCG iterates of the two solves.
    Both row formats of the operator: 16-byte factor records (the sweep rebuilds the slots) and dense 27-slot rows."""
    import scipy.sparse as sp
    monkeypatch.setenv('NKSR_ROW_FORMAT', row_format)
    from nksr_amd import solver
    from nksr_amd.fields import KernelField
    from oracle import solve
    xyz, nrm, oh, svh, feats, ointerps, net = _setup(n=3000)
    fld = KernelField(svh, net.interpolators, [torch.from_numpy(f) for f in feats], approx_kernel_grad=approx)
    t = lambda a: torch.from_numpy(a).to(_dev())
    nxyz = np.concatenate([oh.levels[0].centers(), oh.levels[1].centers()])          # normal sites on two levels (adaptive_depth 2)
    nval = np.random.RandomState(5).randn(len(nxyz), 3).astype(np.float32)
    wp, wn = 1e4 / len(xyz), 1e4 / len(nxyz) * 0.01
    A, b, _, _, _ = solve.assemble(oh, feats, ointerps, xyz, nxyz, nval, wp, wn, 1.0, approx)
    rowptr, cols, vals, diag, gb = fld.assemble(t(xyz), t(nxyz), t(nval), wp, wn, 1.0)
    op = fld.fused_operator(t(xyz), t(nxyz), t(nval), wp, wn)
    assert op['row_format'] == row_format
    M = A.shape[0]
    fb, fd = fld.fused_rhs_diag(op, 1.0)
    pu.check('fused:rhs_rel', np.abs(fb.cpu().numpy() - b).max() / np.abs(b).max(), 1e-5)
    pu.check('fused:diag_rel', (np.abs(fd.cpu().numpy() - A.diagonal()) / A.diagonal()).max(), 1e-5)
    pu.check('fused:diag_vs_csr_rel', float(((fd - diag).abs() / diag).max()), 1e-5)
    rs = np.random.RandomState(1)
    A64 = A.astype(np.float64)
    for trial in range(3):
        x = rs.randn(M).astype(np.float32)
        yf = fld.fused_apply(op, t(x)).cpu().numpy()
        yc = solver.spmv(rowptr, cols, vals, t(x)).cpu().numpy()
        bound = abs(A64) @ abs(x).astype(np.float64)                                  # |A||x|: the fp32 rounding scale of one product
        pu.check('fused:apply_vs_oracle[%d]' % trial, (np.abs(yf - A64 @ x.astype(np.float64)) / bound).max(), 1e-5)
        pu.check('fused:apply_vs_csr[%d]' % trial, (np.abs(yf - yc) / bound).max(), 1e-5)
    # same PCG, two operators: iterates after a fixed number of iterations
    fld.solver_config.update({'tol': 0.0, 'max_iter': 6, 'check_every': 2, 'coarse_precond': False})      # Jacobi only: the CSR solve's preconditioner
    x_csr = solver.pcg_solve(rowptr, cols, vals, diag, gb, tol=0.0, max_iter=6, check_every=2)[0]
    fld.solve_fused(t(xyz), t(nxyz), t(nval), wp, wn, 1.0)
    assert fld.solve_info['iters'] == 6 and fld.solve_info['fused']
    pu.check('fused:iterate6_vs_csr', float((fld.alpha - x_csr).abs().max() / x_csr.abs().max()), 1e-5)
    # converged: both reach the tolerance, the fields agree
    fld.solver_config.update({'tol': 1e-6, 'max_iter': 2000, 'check_every': 16})
    fld.solve_fused(t(xyz), t(nxyz), t(nval), wp, wn, 1.0)
    a_f, it_f = fld.alpha.clone(), fld.solve_info['iters']
    assert fld.solve_info['rel_residual'] <= 1e-6
    fld.solve_non_fused(t(xyz), t(nxyz), t(nval), wp, wn, 1.0)
    assert abs(fld.solve_info['iters'] - it_f) <= max(3, it_f // 10)
    r = b.astype(np.float64) - A64 @ a_f.cpu().numpy().astype(np.float64)
    pu.check('fused:residual_in_oracle_system', np.linalg.norm(r) / np.linalg.norm(b), 1e-5)
    q = t((xyz[:1500] + np.float32(0.02)).astype(np.float32))
    f_csr = fld.evaluate_f(q).value
    fld.alpha = a_f
    f_fused = fld.evaluate_f(q).value
    pu.check('fused:field_vs_csr', float((f_fused - f_csr).abs().max() / f_csr.abs().max()), 1e-4)


def test_row_order_by_rank_passes_equals_the_sorted_merge(monkeypatch):
    """The shared Morton-ordered row list of the two site sets: first rows from two rank passes over the already sorted key lists
    (nksr_rank_sorted) against the radix sort of the concatenated keys + scan + scatter -- rows, row cells and targets bit for bit;
    the primitive itself against torch.searchsorted, both bounds, with runs of equal keys."""
    from nksr_amd.fields import KernelField
    from nksr_amd._lib import call, ptr, stream
    rs = np.random.RandomState(3)
    a = torch.from_numpy(np.sort(rs.randint(0, 5000, 20000)).astype(np.int64)).to(_dev())
    b = torch.from_numpy(np.sort(rs.randint(0, 5000, 7000)).astype(np.int64)).to(_dev())
    for upper in (0, 1):
        out = torch.empty(b.numel(), dtype=torch.int32, device=_dev())
        call('nksr_rank_sorted', ptr(a), a.numel(), ptr(b), b.numel(), upper, ptr(out), stream())
        assert torch.equal(out.long(), torch.searchsorted(a, b, right=bool(upper)))
    xyz, nrm, oh, svh, feats, ointerps, net = _setup(n=3000)
    fld = KernelField(svh, net.interpolators, [torch.from_numpy(f) for f in feats])
    t = lambda x: torch.from_numpy(x).to(_dev())
    nxyz = np.concatenate([oh.levels[0].centers(), oh.levels[1].centers()])
    nval = np.random.RandomState(5).randn(len(nxyz), 3).astype(np.float32)
    out = {}
    for mode in ('merge', 'sort'):
        monkeypatch.setenv('NKSR_ROW_ORDER', mode)
        op = fld.fused_operator(t(xyz), t(nxyz), t(nval), 1e4 / len(xyz), 1e2 / len(nxyz))
        if op['row_format'] == 'factors':
            n = op['op'].depth * op['rows_total'] * 4
            rows = torch.cat([op['fac_vec'][:n], op['fac_pos'][:op['rows_total'] * 4]])
        else:
            rows = fld.dense_rows(op).reshape(-1)
        out[mode] = (rows.clone(), op['row_cells'].clone(), op['targets_all'].clone(), op['rows_total'])
    assert out['merge'][3] == out['sort'][3]
    assert torch.equal(out['merge'][1], out['sort'][1]) and torch.equal(out['merge'][2], out['sort'][2])
    assert torch.equal(out['merge'][0].view(torch.int32), out['sort'][0].view(torch.int32))


@pytest.mark.parametrize('approx', [False, True])
@pytest.mark.parametrize('hidden', [16, 32])
def test_merged_rows_equal_the_rows_of_a_launch_per_set_bit_for_bit(approx, hidden, monkeypatch):
    """nksr_kernel_rows_merged (one lane per ROW of the operator's merged row list, value + one tangent channel as packed pairs, rows
    leaving as contiguous wavefront images) against nksr_kernel_rows with a row index (one lane per site, a launch per site set):
    rows, row cells and targets bit for bit -- exact and approximate gradients, both interpolator widths, one site set alone; in the
    dense layout and in the COMPACT one (only the slots of a cell's existing neighbours are stored: expanded through the operator's
    tables, the absent slots must be the zeros the dense rows hold there).  The operator and its set-up sums are the same bits in
    both layouts."""
    from nksr_amd.fields import KernelField
    xyz, nrm, oh, svh, feats, ointerps, net = _setup(n=3000, init_scale=0.3, H=hidden)
    fld = KernelField(svh, net.interpolators, [torch.from_numpy(f) for f in feats], approx_kernel_grad=approx)
    t = lambda x: torch.from_numpy(x).to(_dev())
    nxyz = np.concatenate([oh.levels[0].centers(), oh.levels[1].centers()])
    nval = np.random.RandomState(5).randn(len(nxyz), 3).astype(np.float32)
    far = (xyz[:40] + np.float32(3.0)).astype(np.float32)                      # sites outside every cell of the fine levels: zero rows, cell -1
    xv = torch.randn(svh.num_unknowns, device=_dev(), generator=torch.Generator(device=_dev()).manual_seed(3))
    for pos, nrm_sites in ((np.concatenate([xyz, far]), nxyz), (xyz, None), (None, nxyz)):
        out = {}
        for mode, layout in (('site', 'dense'), ('merged', 'dense'), ('merged', 'compact')):
            monkeypatch.setenv('NKSR_ROWS_KERNEL', mode)
            monkeypatch.setenv('NKSR_ROWS_LAYOUT', layout)
            op = fld.fused_operator(t(pos) if pos is not None else None, t(nrm_sites) if nrm_sites is not None else None,
                                    t(nval) if nrm_sites is not None else None, 1e4 / 3000, 1e2 / len(nxyz))
            assert op['row_format'] == 'dense' and op['compact'] == (layout == 'compact')
            b, dg = fld.fused_rhs_diag(op, 1.0)
            out[(mode, layout)] = (fld.dense_rows(op).clone(), op['row_cells'].clone(), op['targets_all'].clone(), b, dg, fld.fused_apply(op, xv), op)
        ref = out[('site', 'dense')]
        for key in (('merged', 'dense'), ('merged', 'compact')):
            o = out[key]
            assert torch.equal(ref[1], o[1]) and torch.equal(ref[2], o[2]), key
            assert torch.equal(ref[0].view(torch.int32), o[0].view(torch.int32)), key
            assert torch.equal(ref[3], o[3]) and torch.equal(ref[4], o[4]) and torch.equal(ref[5], o[5]), key      # rhs, diagonal, A x: same bits
        cop = out[('merged', 'compact')][6]
        assert float(ref[0].abs().max()) > 0
        if pos is not None and nrm_sites is not None:
            assert cop['rows_words'] < ref[0].numel()      # (smaller: absent neighbours; a set of one-row cells can lose that to the 16-byte padding)
        # the coarse-level block of the preconditioner from compact rows == from dense rows
        if pos is not None and nrm_sites is not None:
            blk = {}
            for key in (('merged', 'dense'), ('merged', 'compact')):
                rp, cc, vv, dd, _ = fld.assemble(None, None, None, 1.0, 1.0, 1.0, coarse_from=2, fused_op=out[key][6])
                blk[key] = (rp, cc, vv, dd)
            for a_, b_ in zip(blk[('merged', 'dense')], blk[('merged', 'compact')]):
                assert torch.equal(a_, b_)


@pytest.mark.parametrize('fused', [False, True])
def test_solve_is_differentiable_wrt_the_normal_targets(fused):
    """SURVEY.md section 8(f)-4, the part that is built: under autograd, solve*() makes alpha a differentiable function of
    ``normal_value`` (models/nksr_net.py:105-112 passes the normal head's output there) by implicit differentiation -- one more PCG
    solve with the same system -- and evaluate_f is differentiable in alpha (transposed pass of the matrix-free operator).
    Checked against EXACT directional derivatives from the oracle: alpha is linear in the targets, so a loss that is linear /
    quadratic in (f, grad f) has exact first / central differences."""
    import scipy.sparse.linalg as sla
    from nksr_amd.fields import KernelField
    from oracle import field as ofield, solve
    xyz, nrm, oh, svh, feats, ointerps, net = _setup(n=1500, init_scale=0.3)
    fld = KernelField(svh, net.interpolators, [torch.from_numpy(f) for f in feats])
    fld.solver_config.update({'tol': 1e-7, 'max_iter': 4000})
    t = lambda a: torch.from_numpy(a).to(_dev())
    nxyz = oh.levels[0].centers()
    rs = np.random.RandomState(3)
    nval = rs.randn(len(nxyz), 3).astype(np.float32)
    wp, wn = 1e4 / len(xyz), 1e4 / len(nxyz) * 0.01
    A, b, G, Q, psis = solve.assemble(oh, feats, ointerps, xyz, nxyz, nval, wp, wn, 1.0)
    lu = sla.splu(A.astype(np.float64).tocsc())
    q = (xyz[:400] + rs.randn(400, 3).astype(np.float32) * np.float32(0.02)).astype(np.float32)
    c1, c3 = rs.randn(400), rs.randn(400, 3)

    def oracle_loss(nv):
        rhs = wn * (Q.astype(np.float64).T @ np.concatenate([nv[:, a] for a in range(3)]).astype(np.float64))
        alpha = lu.solve(rhs).astype(np.float32)
        f, g = ofield.evaluate_f(oh, feats, ointerps, psis, alpha, q, True, False)
        return float((c1 * f).sum() + (c3 * g).sum() + 0.5 * (f.astype(np.float64) ** 2).sum())

    nv_t = t(nval).requires_grad_(True)
    (fld.solve if fused else fld.solve_non_fused)(t(xyz), t(nxyz), nv_t, wp, wn, 1.0)
    assert fld.alpha.requires_grad and bool(fld.solve_info.get('fused', False)) == fused
    res = fld.evaluate_f(t(q), grad=True)
    loss = (t(c1.astype(np.float32)) * res.value).sum() + (t(c3.astype(np.float32)) * res.gradient).sum() + 0.5 * (res.value ** 2).sum()
    pu.check('autograd[fused=%s]:loss_rel' % fused, abs(float(loss) - oracle_loss(nval)) / abs(oracle_loss(nval)), 1e-4)
    loss.backward()
    g = nv_t.grad.cpu().numpy().astype(np.float64)
    assert g.shape == nval.shape and np.isfinite(g).all() and np.abs(g).max() > 0
    for trial in range(3):                # exact central differences of the (quadratic) loss along random directions
        v = rs.randn(*nval.shape).astype(np.float32)
        fd = 0.5 * (oracle_loss(nval + v) - oracle_loss(nval - v))
        pu.check('autograd[fused=%s]:directional_derivative[%d]' % (fused, trial), abs((g * v).sum() - fd) / max(abs(fd), 1e-12), 2e-3)
    # no autograd, no graph: the plain solve keeps returning a leaf
    with torch.no_grad():
        fld.solve_non_fused(t(xyz), t(nxyz), t(nval), wp, wn, 1.0)
    assert not fld.alpha.requires_grad


@pytest.mark.parametrize('approx', [False, True])
def test_torch_rows_match_the_hip_rows(approx):
    """fields/kernel_rows_torch.py (the differentiable statement of the kernel rows the training-path backward goes through)
    against nksr_kernel_rows: value rows and gradient rows, exact and approx_kernel_grad, incl. sites outside the finest level."""
    from nksr_amd.fields import KernelField, kernel_rows_torch as krt
    xyz, nrm, oh, svh, feats, ointerps, net = _setup(n=2500, init_scale=0.4)
    net.to(_dev())
    tf = [torch.from_numpy(f).to(_dev()) for f in feats]
    fld = KernelField(svh, net.interpolators, tf, approx_kernel_grad=approx)
    q = np.concatenate([xyz[:900], oh.levels[0].centers()[:300], oh.levels[2].centers()[:50] + np.float32(0.3)]).astype(np.float32)
    qt = torch.from_numpy(q).to(_dev())
    hv, hd = fld.kernel_rows(qt, grad=True)
    with torch.no_grad():
        tv, idx = krt.rows(svh, net.interpolators, tf, qt, False, approx)
        td, _ = krt.rows(svh, net.interpolators, tf, qt, True, approx, scale=0.7)
    pu.check('torch_rows[approx=%s]:val' % approx, float((tv - hv).abs().max() / hv.abs().max()), 1e-5)
    pu.check('torch_rows[approx=%s]:dval' % approx, float((td / 0.7 - hd).abs().max() / hd.abs().max()), 1e-5)
    # the index table: global unknown index of every slot
    off = svh.offsets
    for d in range(svh.depth):
        cell = svh.level(d).hash.query(_level_keys(svh, qt, d))
        nb = svh.level(d).nbr[cell.clamp(min=0).long()].long()
        want = torch.where((cell >= 0)[:, None] & (nb >= 0), nb + off[d], torch.full_like(nb, -1))
        assert torch.equal(idx[:, d], want)


def _level_keys(svh, xyz, d):
    from nksr_amd._lib import call, ptr, stream
    p = xyz * torch.tensor(svh.inv_w0, dtype=torch.float32, device=xyz.device)
    ijk = ((torch.floor(p * 2.0).long() >> d) >> 1).to(torch.int32).contiguous()
    keys = torch.empty(xyz.shape[0], dtype=torch.int64, device=xyz.device)
    call('nksr_encode_keys', ptr(ijk), xyz.shape[0], d, ptr(keys), stream())
    return keys


def test_training_path_forward_and_backward_share_their_support():
    """ADVICE.md (round 2): evaluate_f looks the neighbours of a query OUTSIDE every active cell up in the hash, the kernel rows the
    backward differentiates exist only inside active cells -- for samples near the support boundary dL/dalpha disagreed with
    the forward.  Under autograd the forward now uses the rows' support: f is linear in alpha, so dL/dalpha . v must equal
    L(alpha + v) - L(alpha) for queries just outside the active voxels; an empty query batch gives zero gradients."""
    from nksr_amd.fields import KernelField
    xyz, nrm, oh, svh, feats, ointerps, net = _setup(n=1500, init_scale=0.3)
    net.to(_dev())
    t = lambda a: torch.from_numpy(a).to(_dev())
    fld = KernelField(svh, net.interpolators, [t(f) for f in feats], approx_kernel_grad=False)
    rs = np.random.RandomState(5)
    # queries: input points pushed outwards by 0.5 .. 2.5 finest voxels along their normals (many leave the finest level's cells)
    q = (xyz[:800] + nrm[:800] * (rs.uniform(0.5, 2.5, (800, 1)) * oh.voxel_size).astype(np.float32)).astype(np.float32)
    qt = t(q)
    outside0 = svh.level(0).hash.query(_level_keys(svh, qt, 0)) < 0
    assert int(outside0.sum()) > 50, 'the probe must contain queries outside the finest level'
    a0 = t(rs.randn(svh.num_unknowns).astype(np.float32))
    v = t(rs.randn(svh.num_unknowns).astype(np.float32))
    c1, c3 = t(rs.randn(800).astype(np.float32)), t(rs.randn(800, 3).astype(np.float32))

    def loss_of(alpha):
        fld.alpha = alpha
        r = fld.evaluate_f(qt, grad=True)
        return (c1 * r.value).sum() + (c3 * r.gradient).sum()
    with torch.enable_grad():
        al = a0.clone().requires_grad_(True)
        L0 = loss_of(al)
        g, = torch.autograd.grad(L0, al)
        L1 = loss_of((a0 + v).requires_grad_(True))
        # no queries: zero gradient, no exception
        fld.alpha = al
        e = fld.evaluate_f(torch.zeros((0, 3), device=_dev()), grad=True)
        ge, = torch.autograd.grad(e.value.sum() + e.gradient.sum() + 0.0 * al.sum(), al, allow_unused=True)
    assert ge is None or float(ge.abs().max()) == 0.0
    lin, dif = float((g * v).sum()), float(L1 - L0)
    pu.check('training_support:dL/dalpha.v vs L(alpha+v)-L(alpha)', abs(lin - dif) / max(abs(dif), 1e-12), 2e-4)
    # the inference path still sees the fallback neighbours: on this probe it differs from the training forward
    with torch.no_grad():
        fld.alpha = a0
        inf = fld.evaluate_f(qt).value
    with torch.enable_grad():
        fld.alpha = a0.clone().requires_grad_(True)
        trn = fld.evaluate_f(qt).value.detach()
    d = (inf - trn).abs()
    assert float(d.max()) > 0


@pytest.mark.parametrize('fused', [False, True])
def test_adjoint_solve_takes_the_forward_preconditioner(fused):
    """ADVICE.md (round 2): the backward pass's A^-1 g ignored the coarse-level block the forward solve had used (4x the
    iterations at depth 5).  With the block the adjoint solve gives the same vector as with Jacobi alone."""
    from nksr_amd.fields import KernelField
    xyz, nrm, oh, svh, feats, ointerps, net = _setup(n=1200, init_scale=0.3)
    net.to(_dev())
    t = lambda a: torch.from_numpy(a).to(_dev())
    fld = KernelField(svh, net.interpolators, [t(f) for f in feats], approx_kernel_grad=True)
    fld.solver_config.update({'tol': 1e-7, 'max_iter': 4000, 'coarse_precond': {'first_level': max(1, svh.depth - 2)}})
    nxyz = oh.levels[0].centers()
    rs = np.random.RandomState(3)
    nval = t(rs.randn(len(nxyz), 3).astype(np.float32)).requires_grad_(True)
    wp, wn = 1e4 / len(xyz), 1e4 / len(nxyz) * 0.01
    with torch.enable_grad():
        (fld.solve if fused else fld.solve_non_fused)(t(xyz), t(nxyz), nval, wp, wn, 1.0)
    assert fld._pc is not None and fld.solve_info['coarse_precond'] is not None
    g = t(rs.randn(svh.num_unknowns).astype(np.float32))
    with_block = fld._solve_system(g)
    pc, fld._pc = fld._pc, None
    jacobi = fld._solve_system(g)
    fld._pc = pc
    pu.check('adjoint_precond[fused=%s]' % fused, float((with_block - jacobi).abs().max() / jacobi.abs().max()), 1e-4)
    # and the gradient through it is the one of the Jacobi-only adjoint
    loss = (fld.alpha * g).sum()
    ga = torch.autograd.grad(loss, nval, retain_graph=True)[0]
    fld._pc = None
    gb = torch.autograd.grad(loss, nval)[0]
    pu.check('adjoint_precond[fused=%s]:grad' % fused, float((ga - gb).abs().max() / gb.abs().max()), 1e-4)


@pytest.mark.parametrize('approx,fused', [(False, False), (True, False), (True, True)])
def test_solve_is_differentiable_wrt_features_and_interpolators(approx, fused):
    """models/nksr_net.py:105-112 back-propagates through solve_non_fused into the basis features and the
    interpolator weights.  alpha(theta) by implicit differentiation (one more PCG solve), the theta terms as the vector-Jacobian
    product through the torch statement of the rows; checked against central differences of the ORACLE's loss (assembly, direct
    solve and evaluation redone for theta +- eps v) along random directions in feature space and in weight space."""
    import scipy.sparse.linalg as sla
    from nksr_amd.fields import KernelField
    from oracle import field as ofield, kernel, solve
    xyz, nrm, oh, svh, feats, ointerps, net = _setup(n=1200, init_scale=0.3)
    net.to(_dev())
    t = lambda a: torch.from_numpy(a).to(_dev())
    tf = [t(f).requires_grad_(True) for f in feats]
    fld = KernelField(svh, net.interpolators, tf, approx_kernel_grad=approx)
    fld.solver_config.update({'tol': 1e-7, 'max_iter': 4000})
    nxyz = oh.levels[0].centers()
    rs = np.random.RandomState(7)
    nval = rs.randn(len(nxyz), 3).astype(np.float32)
    wp, wn = 1e4 / len(xyz), 1e4 / len(nxyz) * 0.01
    q = (xyz[:300] + rs.randn(300, 3).astype(np.float32) * np.float32(0.02)).astype(np.float32)
    c1, c3 = rs.randn(300), rs.randn(300, 3)

    def oracle_loss(fs, its):
        A, b, G, Q, psis = solve.assemble(oh, fs, its, xyz, nxyz, nval, wp, wn, 1.0, approx)
        alpha = sla.splu(A.astype(np.float64).tocsc()).solve(b.astype(np.float64)).astype(np.float32)
        f, g = ofield.evaluate_f(oh, fs, its, psis, alpha, q, True, approx)
        return float((c1 * f).sum() + (c3 * g).sum() + 0.5 * (f.astype(np.float64) ** 2).sum())

    (fld.solve if fused else fld.solve_non_fused)(t(xyz), t(nxyz), t(nval), wp, wn, 1.0)
    assert fld.alpha.requires_grad and bool(fld.solve_info.get('fused', False)) == fused
    res = fld.evaluate_f(t(q), grad=True)
    loss = (t(c1.astype(np.float32)) * res.value).sum() + (t(c3.astype(np.float32)) * res.gradient).sum() + 0.5 * (res.value ** 2).sum()
    base = oracle_loss(feats, ointerps)
    pu.check('autograd_theta[approx=%s]:loss_rel' % approx, abs(float(loss) - base) / abs(base), 1e-4)
    params = [p for it in net.interpolators for p in (it.W1, it.b1, it.W2, it.b2, it.W3, it.b3)]
    grads = torch.autograd.grad(loss, tf + params)
    gf = [g.cpu().numpy().astype(np.float64) for g in grads[:len(tf)]]
    gp = [g.cpu().numpy().astype(np.float64) for g in grads[len(tf):]]
    assert all(np.isfinite(g).all() for g in gf + gp) and max(np.abs(g).max() for g in gf) > 0 and max(np.abs(g).max() for g in gp) > 0
    # (1) the implicit-function algebra, independent of finite differences: the same loss through a DENSE float64 solve of the
    #     system built from the torch rows, differentiated end to end by autograd (exact-gradient rows are discontinuous in theta
    #     across ReLU kinks -- d(phi)/dx is piecewise constant -- so central differences are only meaningful in approx mode)
    from nksr_amd.fields import kernel_rows_torch as krt
    M = svh.num_unknowns

    def dense(R, idx, rows_per_site):
        n = R.shape[0]
        Rf = R.reshape(n * rows_per_site, -1) if rows_per_site == 1 else R.reshape(n, 3, -1).reshape(n * 3, -1)
        ix = idx.reshape(n, -1)
        ix = ix if rows_per_site == 1 else ix[:, None, :].expand(n, 3, ix.shape[1]).reshape(n * 3, -1)
        D = torch.zeros((Rf.shape[0], M + 1), dtype=torch.float64, device=R.device)
        D = D.scatter_add(1, torch.where(ix >= 0, ix, torch.full_like(ix, M)), Rf.double())
        return D[:, :M]
    RG, iG = krt.rows(svh, net.interpolators, tf, t(xyz), False, approx, scale=wp ** 0.5)
    RQ, iQ = krt.rows(svh, net.interpolators, tf, t(nxyz), True, approx, scale=wn ** 0.5)
    Gd, Qd = dense(RG, iG, 1), dense(RQ, iQ, 3)
    Ad = Gd.T @ Gd + Qd.T @ Qd + torch.eye(M, dtype=torch.float64, device=_dev())
    bd = Qd.T @ (t(nval).double().reshape(-1) * wn ** 0.5)
    ad = torch.linalg.solve(Ad, bd)
    Rf_, if_ = krt.rows(svh, net.interpolators, tf, t(q), False, approx)
    Rg_, ig_ = krt.rows(svh, net.interpolators, tf, t(q), True, approx)
    fd_ = dense(Rf_, if_, 1) @ ad
    gd_ = (dense(Rg_, ig_, 3) @ ad).reshape(-1, 3)
    loss_d = (t(c1).double() * fd_).sum() + (t(c3).double() * gd_).sum() + 0.5 * (fd_ ** 2).sum()
    pu.check('autograd_theta[approx=%s]:dense_loss_rel' % approx, abs(float(loss_d) - float(loss)) / abs(float(loss)), 1e-4)
    gd_all = torch.autograd.grad(loss_d, tf + params)
    num = sum(float(((a_.double() - b_.double()) ** 2).sum()) for a_, b_ in zip(grads, gd_all)) ** 0.5
    den = sum(float((b_.double() ** 2).sum()) for b_ in gd_all) ** 0.5
    pu.check('autograd_theta[approx=%s]:gradient_vs_dense_autograd_rel_l2' % approx, num / den, 1e-4)
    for name, lo, hi in (('features', 0, len(tf)), ('weights', len(tf), len(tf) + len(params))):
        num = sum(float(((a_.double() - b_.double()) ** 2).sum()) for a_, b_ in zip(grads[lo:hi], gd_all[lo:hi])) ** 0.5
        den = sum(float((b_.double() ** 2).sum()) for b_ in gd_all[lo:hi]) ** 0.5
        pu.check('autograd_theta[approx=%s]:%s_vs_dense_autograd_rel_l2' % (approx, name), num / den, 1e-4)
    if not approx or fused:
        return
    # (2) approx mode (rows continuous in theta): central differences of the ORACLE's loss along random directions
    eps = 2e-3
    for trial in range(1):            # a direction in feature space (two oracle solves each: the slowest part of the GPU suite on a slow host)
        v = [rs.randn(*f.shape).astype(np.float32) for f in feats]
        fd = (oracle_loss([f + np.float32(eps) * d for f, d in zip(feats, v)], ointerps) -
              oracle_loss([f - np.float32(eps) * d for f, d in zip(feats, v)], ointerps)) / (2 * eps)
        an = sum((g * d).sum() for g, d in zip(gf, v))
        pu.report('autograd_theta[approx=%s]:d_features[%d]:values' % (approx, trial), analytic=float(an), finite_difference=float(fd), loss=float(base))
        pu.check('autograd_theta[approx=%s]:d_features[%d]' % (approx, trial), abs(an - fd) / max(abs(fd), 1e-12), 6e-2)
    P0 = [p.detach().cpu().numpy() for p in params]
    for trial in range(1):            # a direction in weight space
        v = [rs.randn(*p.shape).astype(np.float32) for p in P0]

        def interps(sign):
            P = [p + np.float32(sign * eps) * d for p, d in zip(P0, v)]
            return [kernel.Interpolator(*P[6 * i:6 * i + 6]) for i in range(len(net.interpolators))]
        fd = (oracle_loss(feats, interps(1.0)) - oracle_loss(feats, interps(-1.0))) / (2 * eps)
        an = sum((g * d).sum() for g, d in zip(gp, v))
        pu.report('autograd_theta[approx=%s]:d_weights[%d]:values' % (approx, trial), analytic=float(an), finite_difference=float(fd), loss=float(base))
        pu.check('autograd_theta[approx=%s]:d_weights[%d]' % (approx, trial), abs(an - fd) / max(abs(fd), 1e-12), 6e-2)


@pytest.mark.parametrize('K,H', [(4, 16), (16, 32)])
@pytest.mark.parametrize('approx', [False, True])
def test_hip_theta_vjp_matches_autograd_through_the_torch_rows(K, H, approx, monkeypatch):
    """The theta term of the solve's / evaluate_f's backward as HIP kernels (nksr_kernel_rows_vjp + nksr_voxel_psi_vjp: one thread
    per (site, level) recomputes its row and pushes the cotangents into the basis features, the neighbours' psi and the
    interpolator weights) against torch autograd through fields/kernel_rows_torch.py, the differentiable statement of the same
    rows: value rows and gradient rows (exact and approx_kernel_grad), with and without the lambda term, both (kernel_dim,
    hidden) pairs of the presets.  Relative L2 error per parameter group; the HIP path must not touch the torch rows."""
    from nksr_amd.fields import KernelField, kernel_rows_torch as krt
    xyz, nrm, oh, svh, feats, ointerps, net = _setup(n=1500, K=K, H=H, init_scale=0.3)
    net.to(_dev())
    t = lambda a: torch.from_numpy(a).to(_dev())
    tf = [t(f).requires_grad_(True) for f in feats]
    fld = KernelField(svh, net.interpolators, tf, approx_kernel_grad=approx)
    rs = np.random.RandomState(11)
    M = svh.num_unknowns
    alpha, lam = t(rs.randn(M).astype(np.float32)), t(rs.randn(M).astype(np.float32))
    q = t((xyz[:400] + rs.randn(400, 3).astype(np.float32) * np.float32(0.03)).astype(np.float32))       # some queries leave the finest level
    nx = t(oh.levels[0].centers()[:500].astype(np.float32))
    tn = t(rs.randn(500, 3).astype(np.float32))
    g1, g3 = t(rs.randn(400).astype(np.float32)), t(rs.randn(400, 3).astype(np.float32))
    cases = {'solve': ([(q, False, 0.7, lambda u, v: (-u, -v)), (nx, True, 1.3, lambda u, v: (tn - u, -v))], lam),
             'evaluate': ([(q, False, 1.0, lambda u, v: (None, g1)), (q, True, 1.0, lambda u, v: (None, g3))], None)}
    for name, (sets, lm) in cases.items():
        monkeypatch.setenv('NKSR_THETA_VJP', 'torch')
        ref = fld._theta_vjp(sets, alpha, lm)
        monkeypatch.setenv('NKSR_THETA_VJP', 'hip')
        with monkeypatch.context() as mp:
            mp.setattr(krt, 'rows', lambda *a, **k: (_ for _ in ()).throw(AssertionError('the HIP path went through the torch rows')))
            got = fld._theta_vjp(sets, alpha, lm)
        assert len(got) == len(ref) == len(tf) + 6 * len(net.interpolators)
        for grp, lo, hi in (('features', 0, len(tf)), ('weights', len(tf), len(ref))):
            num = sum(float(((a.double() - b.double()) ** 2).sum()) for a, b in zip(got[lo:hi], ref[lo:hi])) ** 0.5
            den = sum(float((b.double() ** 2).sum()) for b in ref[lo:hi]) ** 0.5
            assert den > 0
            pu.check('theta_vjp_hip[K=%d,H=%d,approx=%s]:%s:%s_rel_l2' % (K, H, approx, name, grp), num / den, 2e-5)
        assert all(a.shape == b.shape and torch.isfinite(a).all() for a, b in zip(got, ref))


def test_pcg_matches_oracle_and_scipy():
    import scipy.sparse as sp
    import scipy.sparse.linalg as sla
    from nksr_amd import solver
    from nksr_amd.fields import KernelField
    from oracle import solve
    xyz, nrm, oh, svh, feats, ointerps, net = _setup(n=3000, random_feats=False, init_scale=0.0)
    fld = KernelField(svh, net.interpolators, [torch.from_numpy(f) for f in feats])
    t = lambda a: torch.from_numpy(a).to(_dev())
    nxyz = oh.levels[0].centers()
    nval = np.random.RandomState(5).randn(len(nxyz), 3).astype(np.float32)
    rowptr, cols, vals, diag, b = fld.assemble(t(xyz), t(nxyz), t(nval), 1e4 / len(xyz), 1e4 / len(nxyz) * 0.01, 1.0)
    M = b.numel()
    lc, lv = solver.csr_logical(rowptr, cols, vals)
    Ag = sp.csr_matrix((lv.cpu().numpy(), lc.cpu().numpy(), rowptr.cpu().numpy()), shape=(M, M))
    bn = b.cpu().numpy()
    # (1) few fixed iterations: iterates agree tightly (same matrix, same algorithm)
    x5, it5, _ = solver.pcg_solve(rowptr, cols, vals, diag, b, tol=0.0, max_iter=6, check_every=2)
    o5, _, _ = solve.pcg_jacobi(Ag, bn, fixed_iters=6)
    assert it5 == 6
    np.testing.assert_allclose(x5.cpu().numpy(), o5, rtol=0, atol=1e-4 * abs(o5).max())
    # (2) converged: residual below tol, solution matches fp64 scipy CG
    x, it, rel = solver.pcg_solve(rowptr, cols, vals, diag, b, tol=1e-6, max_iter=2000, check_every=7)
    xo, ito, relo = solve.pcg_jacobi(Ag, bn, tol=1e-6)
    assert rel <= 1e-6 and abs(it - ito) <= max(3, ito // 10), (it, ito)
    r = bn - Ag.astype(np.float64) @ x.cpu().numpy().astype(np.float64)
    assert np.linalg.norm(r) / np.linalg.norm(bn) <= 2e-6
    xs, info = sla.cg(Ag.astype(np.float64), bn.astype(np.float64), rtol=1e-12, maxiter=20000)
    assert info == 0
    assert abs(x.cpu().numpy() - xs).max() <= 1e-3 * abs(xs).max()
    # zero right-hand side: zero iterations, zero solution
    x0, it0, _ = solver.pcg_solve(rowptr, cols, vals, diag, torch.zeros_like(b))
    assert it0 == 0 and float(x0.abs().max()) == 0.0


@pytest.mark.parametrize('approx', [False, True])
def test_evaluate_f(approx):
    from nksr_amd.fields import KernelField
    from oracle import field, kernel
    xyz, nrm, oh, svh, feats, ointerps, net = _setup(n=2000)
    fld = KernelField(svh, net.interpolators, [torch.from_numpy(f) for f in feats], approx_kernel_grad=approx)
    alpha = np.random.RandomState(2).randn(oh.num_unknowns).astype(np.float32)
    fld.alpha = torch.from_numpy(alpha).to(_dev())
    psis = [kernel.voxel_psi(feats[d], ointerps[d]) for d in range(4)]
    rs = np.random.RandomState(4)
    q = np.concatenate([xyz[:800] + rs.randn(800, 3).astype(np.float32) * 0.03, (rs.rand(200, 3).astype(np.float32) - 0.5) * 3])
    fo, go = field.evaluate_f(oh, feats, ointerps, psis, alpha, q, True, approx)
    res = fld.evaluate_f(torch.from_numpy(q).to(_dev()), grad=True)
    mag = abs(fo).max()
    np.testing.assert_allclose(res.value.cpu().numpy(), fo, rtol=0, atol=1e-4 * mag)
    np.testing.assert_allclose(res.gradient.cpu().numpy(), go, rtol=0, atol=1e-4 * abs(go).max())
    # occupancy convention of models/loss.py:99: evaluate_f_bar(x) > 0 <=> inside; same numbers as evaluate_f
    assert torch.equal(fld.evaluate_f_bar(torch.from_numpy(q).to(_dev())), fld.evaluate_f(torch.from_numpy(q).to(_dev())).value)


@pytest.mark.parametrize('kind,vs', [('sphere', 0.05), ('torus', 0.04)])
def test_end_to_end_mesh(kind, vs):
    """Full reconstruct -> extract_dual_mesh against the oracle pipeline; topology index-exact
    wherever |f| at the lattice vertices is away from the sign threshold."""
    import nksr_amd
    from oracle import pipeline
    xyz, nrm = make_cloud(kind, 3000, 0.005, 0)
    rec = nksr_amd.Reconstructor(_dev())
    # sphere: the default matrix-free solve (fused_mode=True); torus: the assembled CSR solve
    fld = rec.reconstruct(torch.from_numpy(xyz).to(_dev()), torch.from_numpy(nrm).to(_dev()), voxel_size=vs, solver_tol=1e-6,
                          fused_mode=(kind == 'sphere'))
    assert bool(fld.solve_info.get('fused', False)) == (kind == 'sphere')
    scale = 0.1 / vs
    ofl = pipeline.reconstruct((xyz * np.float32(scale)).astype(np.float32), nrm, tol=1e-6)
    assert fld.solve_info['M'] == ofl['A'].shape[0]
    # field values at the input points agree (both solved to 1e-6)
    fo, _ = pipeline.evaluate(ofl, (xyz * np.float32(scale)).astype(np.float32))
    fg = fld.evaluate_f(torch.from_numpy(xyz).to(_dev())).value.cpu().numpy()
    pu.check('e2e[%s]:f_at_inputs' % kind, np.abs(fg - fo).max() / np.abs(ofl['alpha']).max(), 1e-4)
    pu.check_alpha('e2e[%s]' % kind, fld.alpha.cpu().numpy(), ofl, 1e-6)
    for mise in (0, 1, 2):
        # topology: index-exact outside the near-threshold cells, vertices within 1e-4 voxel (tests/parity_util.py)
        st, mesh, _ = pu.mesh_parity('e2e[%s,mise=%d]' % (kind, mise), fld, ofl, mise, scale)
        gv, gf = mesh.v.cpu().numpy() * scale, mesh.f.cpu().numpy()
        E = pu.assert_closed(gf, 'mesh')          # closed at every MISE level (hanging-vertex constraint)
        assert len(gv) - E + len(gf) == (2 if kind == 'sphere' else 0)


@pytest.mark.parametrize('name', ['bunny_2k', 'sphere_3k'])
def test_against_committed_golden(name):
    """HIP path vs the committed oracle fixtures (tests/golden, oracle/make_golden.py)."""
    import os
    import nksr_amd
    from nksr_amd import utils
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    g = np.load(os.path.join(gold, name + '_golden.npz'))
    if name == 'bunny_2k':
        d = np.load(os.path.join(gold, 'bunny_2k.npz'))
        xyz, nrm = d['xyz'], d['normal']
    else:
        xyz, nrm = utils.synth_sphere(3000, 0.45, 0.005, seed=0)
    vs = float(g['voxel_size'])
    rec = nksr_amd.Reconstructor(_dev())
    fld = rec.reconstruct(torch.from_numpy(xyz).to(_dev()), torch.from_numpy(nrm).to(_dev()), voxel_size=vs, solver_tol=1e-6)
    for d in range(4):
        assert np.array_equal(fld.svh.level(d).keys.cpu().numpy(), g['keys_%d' % d])      # voxel sets: exact
    amax = abs(g['alpha']).max()
    pu.check('golden[%s]:alpha_rel' % name, np.abs(fld.alpha.cpu().numpy() - g['alpha']).max() / amax, pu.ALPHA_TOL)
    np.testing.assert_allclose(fld.rhs.cpu().numpy(), g['b'], rtol=1e-4, atol=1e-5 * abs(g['b']).max())
    np.testing.assert_allclose(fld.diag.cpu().numpy(), g['A_diag'], rtol=1e-4)
    res = fld.evaluate_f(torch.from_numpy(xyz).to(_dev()), grad=True)
    pu.check('golden[%s]:f_at_inputs' % name, np.abs(res.value.cpu().numpy() - g['f_at_points']).max() / amax, 1e-4)
    scale = 0.1 / vs
    pu.check('golden[%s]:grad_at_inputs_rel' % name, np.abs(res.gradient.cpu().numpy() / scale - g['grad_at_points']).max()
             / abs(g['grad_at_points']).max(), 1e-4)
    for mise in (0, 1):
        ref = pu.ref_from_golden(g, 'mesh%d_' % mise)
        ev = lambda p: fld._evaluate_f_model(torch.from_numpy(p).to(_dev()), False).value.cpu().numpy()
        delta, fmax = pu.lattice_delta(ev, ref)
        pu.check('golden[%s,mise=%d]:lattice_f_rel' % (name, mise), delta / fmax, 1e-4)
        mesh = fld.extract_dual_mesh(mise_iter=mise)
        pu.compare_meshes('golden[%s,mise=%d]' % (name, mise), *pu.mesh_arrays(mesh, scale), ref, delta_f=delta)


def test_sorted_builders_equal_point_builders():
    """The cell-based hierarchy builders give exactly the voxel sets of the per-point builders."""
    import nksr_amd
    from nksr_amd.nn.network import sort_cloud
    xyz, nrm = make_cloud('torus', 20000, 0.005, 7)
    x = torch.from_numpy((xyz * np.float32(3.0)).astype(np.float32)).to(_dev())
    ks, xs, _ = sort_cloud(x, None, nksr_amd.svh.inv_w0_f32(0.1))
    for depth in (4, 5):
        a = nksr_amd.SparseFeatureHierarchy(0.1, depth, _dev()).build_point_splatting(x)
        b = nksr_amd.SparseFeatureHierarchy(0.1, depth, _dev()).build_point_splatting_sorted(xs, ks)
        c = nksr_amd.SparseFeatureHierarchy(0.1, depth, _dev()).build_point_neighborhood(x)
        d = nksr_amd.SparseFeatureHierarchy(0.1, depth, _dev()).build_point_neighborhood_sorted(ks)
        for lv in range(depth):
            assert torch.equal(a.level(lv).keys, b.level(lv).keys)
            assert torch.equal(c.level(lv).keys, d.level(lv).keys)


def test_field_save_load_roundtrip(tmp_path):
    """Serialised field evaluates and meshes bit-identically after a reload."""
    import nksr_amd
    from nksr_amd import fields
    xyz, nrm = make_cloud('sphere', 3000, 0.005, 0)
    rec = nksr_amd.Reconstructor(_dev())
    fld = rec.reconstruct(torch.from_numpy(xyz).to(_dev()), torch.from_numpy(nrm).to(_dev()), voxel_size=0.05)
    p = str(tmp_path / 'field.pt')
    fields.save_field(fld, p)
    g = fields.load_field(p, _dev())
    q = torch.from_numpy(xyz).to(_dev())
    a, b = fld.evaluate_f(q, grad=True), g.evaluate_f(q, grad=True)
    assert torch.equal(a.value, b.value) and torch.equal(a.gradient, b.gradient)
    m0, m1 = fld.extract_dual_mesh(mise_iter=1), g.extract_dual_mesh(mise_iter=1)
    assert torch.equal(m0.f, m1.f) and torch.equal(m0.v, m1.v)


def test_tree_depth_5_matches_oracle():
    """BASELINE.json configs[4] uses tree_depth=5 (non-default): 135 entries per site row exercise the
    3-column block kernel and the 3-pass row fill."""
    import nksr_amd
    from nksr_amd import configs
    from oracle import network as onet, pipeline
    xyz, nrm = make_cloud('sphere', 4000, 0.003, 0)
    hp = configs.get_hparams('ks', tree_depth=5)
    rec = nksr_amd.Reconstructor(_dev(), hparams=hp)
    fld = rec.reconstruct(torch.from_numpy(xyz).to(_dev()), torch.from_numpy(nrm).to(_dev()), voxel_size=0.03, solver_tol=1e-6)
    xs = (xyz * np.float32(0.1 / 0.03)).astype(np.float32)
    ofl = pipeline.reconstruct(xs, nrm, depth=5, tol=1e-6, net_params=onet.export_params(rec.network))
    assert fld.svh.depth == 5 and fld.solve_info['M'] == ofl['A'].shape[0]
    for d in range(5):
        assert np.array_equal(fld.svh.level(d).keys.cpu().numpy(), ofl['hier'].levels[d].keys)
    ref = np.abs(ofl['alpha']).max()
    pu.check_alpha('depth5', fld.alpha.cpu().numpy(), ofl, 1e-6)
    np.testing.assert_allclose(fld.diag.cpu().numpy(), ofl['A'].diagonal(), rtol=1e-4)
    pu.mesh_parity('depth5[mise=1]', fld, ofl, 1, fld.scale)


def test_coarse_block_preconditioner_same_solution_fewer_iterations():
    """tree_depth 5 (configs[4]): the matrix-free PCG preconditions the levels >= 2 with Chebyshev steps on their diagonal block
    (csrc/pcg.hip, nksr_coarse_precond_t).  Same system, same stopping rule: the solution agrees with the Jacobi-only solve, in
    fewer iterations; the block is the corresponding block of the assembled matrix; two runs are bit-identical."""
    import scipy.sparse as sp
    import nksr_amd
    from nksr_amd import configs, solver
    xyz, nrm = make_cloud('torus', 6000, 0.003, 1)
    hp = configs.get_hparams('ks', tree_depth=5)
    t = lambda a: torch.from_numpy(a).to(_dev())
    out = {}
    for name, pc in (('jacobi', False), ('coarse', None), ('coarse2', None)):
        rec = nksr_amd.Reconstructor(_dev(), hparams=hp)
        rec.coarse_precond = pc
        fld = rec.reconstruct(t(xyz), t(nrm), voxel_size=0.03, solver_tol=1e-6)
        assert fld.solve_info['fused'] and fld.solve_info['rel_residual'] <= 1e-6
        assert (fld.solve_info['coarse_precond'] is None) == (pc is False)
        out[name] = (fld.alpha.clone(), fld.solve_info['iters'], fld.solve_info['coarse_precond'])
    (aj, itj, _), (ac, itc, info), (ac2, itc2, _) = out['jacobi'], out['coarse'], out['coarse2']
    assert torch.equal(ac, ac2) and itc == itc2
    assert info['first_level'] == 2 and info['unknowns'] > 0 and 1.0 < 1.1 * float(info['lambda'].max()) < 100.0
    pu.report('coarse_precond:iters', with_block=itc, jacobi_only=itj)
    assert itc * 2 <= itj, (itc, itj)
    pu.check('coarse_precond:alpha_vs_jacobi_rel', float((ac - aj).abs().max() / aj.abs().max()), pu.ALPHA_TOL)
    # the block itself: rows / columns of the levels >= 2 of the assembled matrix (fp32 Gram sums in another order)
    rec = nksr_amd.Reconstructor(_dev(), hparams=hp)
    f2 = rec.reconstruct(t(xyz), t(nrm), voxel_size=0.03, solver_tol=1e-6, fused_mode=False)
    rowptr, cols, vals, diag = f2.matrix
    M = rowptr.numel() - 1
    lc, lv = solver.csr_logical(rowptr, cols, vals)
    A = sp.csr_matrix((lv.cpu().numpy().astype(np.float64), lc.cpu().numpy(), rowptr.cpu().numpy()), shape=(M, M))
    off = f2.svh.offsets[2]
    # rebuild the operator of the same field and its block through the product path
    svh = f2.svh
    from nksr_amd.nn.network import sort_cloud
    from nksr_amd.svh import inv_w0_f32
    hpp = rec.hparams
    xs = (t(xyz) * f2.scale).contiguous()
    ks, xs, ns = sort_cloud(xs, t(nrm), inv_w0_f32(hpp.voxel_size))
    nxyz = svh.get_voxel_centers(0)
    op = f2.fused_operator(xs, nxyz, torch.zeros_like(nxyz), hpp.solver.pos_weight / xs.shape[0],
                           hpp.solver.normal_weight / nxyz.shape[0] * hpp.voxel_size ** 2, pos_sorted_keys=ks, normal_sorted_keys=svh.level(0).keys)
    rp, cc, vv, dd, _ = f2.assemble(None, None, None, 1.0, 1.0, 1.0, coarse_from=2, fused_op=op)
    n = M - off
    Ac = sp.csr_matrix((vv.cpu().numpy().astype(np.float64), cc.cpu().numpy(), rp.cpu().numpy()), shape=(n, n))
    ref = A[off:, off:]
    D = abs(Ac - ref)
    pu.check('coarse_precond:block_vs_assembled_rel', float(D.max() / abs(ref).max()), 1e-5)
    assert abs(Ac - Ac.T).max() <= 1e-6 * abs(ref).max()


def test_a_coarse_block_that_loses_definiteness_falls_back_to_jacobi_on_the_device():
    """The Chebyshev polynomial of the coarse-level block is only positive definite while its interval covers the block's
    spectrum.  ``lambda_scale`` < 1 forces the bound too small (what a poor power-iteration estimate would do): r.z goes
    <= 0, and instead of raising the PCG restarts that system with Jacobi alone from the current iterate, on the device
    (csrc/pcg.hip: k_spcg_pupdate).  The solve still converges to the Jacobi-only solution; the fallback is counted.  The
    Gershgorin cap never cuts into a sound bound: with the default scale it changes nothing."""
    import nksr_amd
    from nksr_amd import configs
    xyz, nrm = make_cloud('torus', 6000, 0.003, 1)
    hp = configs.get_hparams('ks', tree_depth=5)
    t = lambda a: torch.from_numpy(a).to(_dev())
    out = {}
    for name, pc in (('jacobi', False), ('sound', None), ('broken', {'lambda_scale': 0.25})):
        for fused in (True, False):
            rec = nksr_amd.Reconstructor(_dev(), hparams=hp)
            rec.coarse_precond = pc
            fld = rec.reconstruct(t(xyz), t(nrm), voxel_size=0.03, solver_tol=1e-6, fused_mode=fused)
            assert fld.solve_info['rel_residual'] <= 1e-6, (name, fused, fld.solve_info['rel_residual'])
            out[(name, fused)] = (fld.alpha.clone(), fld.solve_info['iters'], fld.solve_info['jacobi_fallbacks'], fld.solve_info['coarse_precond'])
    for fused in (True, False):
        aj, itj, fbj, _ = out[('jacobi', fused)]
        a_s, its, fbs, info = out[('sound', fused)]
        ab, itb, fbb, _ = out[('broken', fused)]
        assert fbj == 0 and fbs == 0 and fbb == 1, (fused, fbj, fbs, fbb)
        # Gershgorin is a true upper bound of the spectrum the power iteration estimates from below
        assert float(info['lambda'].max()) <= float(info['gershgorin'].max()) * (1 + 1e-5)
        pu.report('coarse_precond:fallback_iters[fused=%s]' % fused, sound=its, broken_then_jacobi=itb, jacobi_only=itj)
        assert its < itb <= itj + 40
        pu.check('coarse_precond:fallback_alpha_vs_jacobi_rel[fused=%s]' % fused, float((ab - aj).abs().max() / aj.abs().max()), pu.ALPHA_TOL)


@pytest.mark.parametrize('n,voxel', [(2500, 0.02), (12000, 0.009)])
def test_shallow_hierarchies_take_the_coarse_block_small_ones_at_once_large_ones_when_jacobi_stalls(n, voxel):
    """Depth-4 single fields: up to 2^16 unknowns the block of the levels >= 1 is taken from the first iteration (round 6: one failed
    Jacobi round was most of a small solve); larger ones start with Jacobi (the dense 1M-point cloud converges in 11 iterations) and
    restart on the residual with the block of the levels >= 2 after one unconverged round of check_every iterations.  Sparse input
    with normal sites on two levels (adaptive_depth 2, the carla preset's) takes 100+ Jacobi iterations either way.  Same solution,
    fewer iterations, deterministic."""
    import nksr_amd
    from nksr_amd import configs
    from nksr_amd.fields.kernel_field import SMALL_FIELD_UNKNOWNS
    k = np.arange(n) + 0.5
    phi, z = np.pi * (1 + 5 ** 0.5) * k, 1 - 2 * k / n
    nrm = np.stack([np.cos(phi) * np.sqrt(1 - z * z), np.sin(phi) * np.sqrt(1 - z * z), z], 1).astype(np.float32)
    xyz = (nrm * np.float32(0.45)).astype(np.float32)
    hp = configs.get_hparams('ks', adaptive_depth=2)
    t = lambda a: torch.from_numpy(a).to(_dev())
    res = {}
    for name, pc in (('jacobi', False), ('auto', None), ('auto2', None)):
        rec = nksr_amd.Reconstructor(_dev(), hparams=hp)
        rec.coarse_precond = pc
        fld = rec.reconstruct(t(xyz), t(nrm), voxel_size=voxel, solver_tol=1e-6)
        assert fld.svh.depth == 4 and fld.solve_info['rel_residual'] <= 1e-6
        res[name] = (fld.alpha.clone(), fld.solve_info['iters'], fld.solve_info['coarse_precond'], fld.solve_info['M'])
    M = res['auto'][3]
    assert (M <= SMALL_FIELD_UNKNOWNS) == (n == 2500), M                                  # one case on either side of the policy's bound
    pu.report('coarse_precond:adaptive_iters[M=%d]' % M, auto=res['auto'][1], jacobi_only=res['jacobi'][1], first_level=res['auto'][2]['first_level'])
    assert res['jacobi'][1] > (48 if n == 2500 else 16) and res['jacobi'][2] is None, res['jacobi'][1]    # (more than one round of check_every: otherwise the input does not exercise the switch)
    assert res['auto'][2] is not None and res['auto'][1] < res['jacobi'][1]
    assert res['auto'][2]['first_level'] == (1 if M <= SMALL_FIELD_UNKNOWNS else 2)
    assert torch.equal(res['auto'][0], res['auto2'][0]) and res['auto'][1] == res['auto2'][1]
    pu.check('coarse_precond:adaptive_alpha_vs_jacobi_rel', float((res['auto'][0] - res['jacobi'][0]).abs().max() / res['jacobi'][0].abs().max()),
             pu.ALPHA_TOL)


def test_adaptive_depth_meshing_covers_what_the_finest_level_leaves_open():
    """adaptive_depth 2 (configs/carla/train.yaml:6; LayerField(dec_svh, adaptive_depth), models/nksr_net.py:132): where the
    finest level is absent (input sparser than the finest voxels) the dual cells of level 1 are meshed too, on the same
    lattice -- the level-0-only mesh has holes, the adaptive one is closed, and it equals the oracle's index for index."""
    import nksr_amd
    from nksr_amd import configs
    from oracle import network as onet, pipeline
    # evenly spaced samples (Fibonacci sphere), spacing ~1.5 finest voxels: too sparse for the level-0 band (+-1 voxel around
    # every sample) to be gap-free, dense enough for level 1
    n = 700
    k = np.arange(n) + 0.5
    phi, z = np.pi * (1 + 5 ** 0.5) * k, 1 - 2 * k / n
    nrm = np.stack([np.cos(phi) * np.sqrt(1 - z * z), np.sin(phi) * np.sqrt(1 - z * z), z], 1).astype(np.float32)
    xyz = (nrm * np.float32(0.45)).astype(np.float32)
    vs = 0.04
    hp = configs.get_hparams('ks', adaptive_depth=2)
    rec = nksr_amd.Reconstructor(_dev(), hparams=hp)
    fld = rec.reconstruct(torch.from_numpy(xyz).to(_dev()), torch.from_numpy(nrm).to(_dev()), voxel_size=vs, solver_tol=1e-6)
    assert fld.meshing_depth == 2
    scale = 0.1 / vs
    ofl = pipeline.reconstruct((xyz * np.float32(scale)).astype(np.float32), nrm, adaptive_depth=2, tol=1e-6,
                               net_params=onet.export_params(rec.network))
    assert fld.solve_info['M'] == ofl['A'].shape[0]

    def open_edges(f):
        e = np.sort(np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]]), 1)
        _, cnt = np.unique(e, axis=0, return_counts=True)
        return int((cnt == 1).sum()), len(cnt)
    for mise in (0, 1):
        st, mesh, _ = pu.mesh_parity('adaptive2[mise=%d]' % mise, fld, ofl, mise, scale)
        gf = mesh.f.cpu().numpy()
        n_open, E = open_edges(gf)
        assert n_open == 0 and mesh.v.shape[0] - E + len(gf) == 2, 'adaptive mesh is not a closed sphere'
    fld.meshing_depth = 1
    n_open0, _ = open_edges(fld.extract_dual_mesh(mise_iter=0).f.cpu().numpy())
    pu.report('adaptive2', open_edges_level0_only=n_open0)
    assert n_open0 > 0, 'the case does not exercise the coarse-level cells'


def test_build_adaptive_normal_variation_matches_oracle():
    """Training-GT hierarchy (models/nksr_net.py:175-179): flat regions stop at a coarse level."""
    from nksr_amd import SparseFeatureHierarchy
    from oracle import hierarchy as oh
    rng = np.random.default_rng(0)
    # a flat plate (no normal variation) next to a small sphere (high variation)
    u = rng.uniform(-1, 1, (6000, 2)).astype(np.float32)
    plate = np.stack([u[:, 0] * 2 + 4, u[:, 1] * 2, np.zeros(6000, np.float32)], 1)
    pn = np.tile(np.array([[0, 0, 1]], np.float32), (6000, 1))
    sph, sn = make_cloud('sphere', 6000, 0.0, 1)
    xyz = np.concatenate([plate, sph * 2]).astype(np.float32)
    nrm = np.concatenate([pn, sn]).astype(np.float32)
    for ad in (1, 2):
        svh = SparseFeatureHierarchy(0.1, 4, _dev())
        svh.build_adaptive_normal_variation(torch.from_numpy(xyz).to(_dev()), torch.from_numpy(nrm).to(_dev()), tau=0.001, adaptive_depth=ad)
        oh_ = oh.Hierarchy(0.1, 4).build_adaptive_normal_variation(xyz, nrm, tau=0.001, adaptive_depth=ad)
        for d in range(4):
            assert np.array_equal(svh.level(d).keys.cpu().numpy(), oh_.levels[d].keys), (ad, d)
        c0 = svh.get_voxel_centers(0).cpu().numpy()
        assert len(c0) > 0 and (c0[:, 0] < 2.5).all()      # nothing fine on the plate
        assert (svh.get_voxel_centers(3).cpu().numpy()[:, 0] > 3).any()


@pytest.mark.parametrize('depth', [1, 2, 3, 6])
def test_other_tree_depths_match_oracle(depth):
    """tree_depth from 1 to NKSR_MAX_DEPTH = 6: exercises every block-tile / row-frame instantiation of the
    assembly (T = 27 .. 162) and the > 64 KB dynamic-LDS path of the structure pass (depth 6)."""
    import nksr_amd
    from nksr_amd import configs
    from oracle import network as onet, pipeline
    xyz, nrm = make_cloud('sphere', 2500, 0.003, depth)
    hp = configs.get_hparams('ks', tree_depth=depth)
    rec = nksr_amd.Reconstructor(_dev(), hparams=hp)
    fld = rec.reconstruct(torch.from_numpy(xyz).to(_dev()), torch.from_numpy(nrm).to(_dev()), voxel_size=0.04, solver_tol=1e-6)
    xs = (xyz * np.float32(0.1 / 0.04)).astype(np.float32)
    ofl = pipeline.reconstruct(xs, nrm, depth=depth, tol=1e-6, net_params=onet.export_params(rec.network))
    assert fld.svh.depth == depth and fld.solve_info['M'] == ofl['A'].shape[0]
    for d in range(depth):
        assert np.array_equal(fld.svh.level(d).keys.cpu().numpy(), ofl['hier'].levels[d].keys)
    np.testing.assert_allclose(fld.diag.cpu().numpy(), ofl['A'].diagonal(), rtol=1e-4)
    ref = np.abs(ofl['alpha']).max()
    pu.check_alpha('depth%d' % depth, fld.alpha.cpu().numpy(), ofl, 1e-6)
    q = (xs[:500] + np.float32(0.03)).astype(np.float32)
    f_gpu = fld._evaluate_f_model(torch.from_numpy(q).to(_dev()), False).value.cpu().numpy()
    f_ref = pipeline.evaluate(ofl, q)[0]
    pu.check('depth%d:f_rel' % depth, np.abs(f_gpu - f_ref).max() / max(np.abs(f_ref).max(), 1e-6), 1e-4)
    pu.mesh_parity('depth%d[mise=1]' % depth, fld, ofl, 1, fld.scale)
