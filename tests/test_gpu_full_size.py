"""BASELINE.json's full single-GPU size (configs[2]: 1 000 000 points, detail_level=1.0): size-independent properties
 and -- what the oracle CAN do at this size -- the voxel hierarchy bit for bit, and kernel rows /
rows of the normal-equation operator against the oracle on a sample of sites / unknowns."""
import hashlib
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def big():
    import nksr_amd
    from nksr_amd import utils
    dev = torch.device('cuda:0')
    xyz, nrm = utils.synth_scene(1_000_000, seed=0)
    xyz, nrm = torch.from_numpy(xyz).to(dev), torch.from_numpy(nrm).to(dev)
    rec = nksr_amd.Reconstructor(dev)
    rec.keep_solve_inputs = True
    fld = rec.reconstruct(xyz, nrm, detail_level=1.0, fused_mode=False)      # the assembled system is inspected below
    return rec, fld, xyz, nrm


def _ranges(starts, ends):
    """concatenated aranges [starts[i], ends[i])"""
    n = (ends - starts).astype(np.int64)
    tot = int(n.sum())
    if tot == 0:
        return np.zeros(0, np.int64)
    off = np.repeat(np.cumsum(n) - n, n)
    return np.repeat(starts.astype(np.int64), n) + (np.arange(tot) - off)


def test_full_size_hierarchy_rows_and_operator_rows_match_the_oracle(big):
    """The headline configuration against the oracle where the oracle can go:
      * the detail_level scale and all four levels of voxel keys, exactly (integer work);
      * the dense-slot kernel rows of 6 000 sampled position sites and 3 000 sampled normal sites (values and gradients);
      * rows of y = (w_p G^T G + w_n Q^T Q + reg I) x for sampled unknowns of every level: the oracle evaluates ALL constraint
        rows in the support of those unknowns (ten thousands of sites for a coarse one) -- against the matrix-free operator
        (nksr_fused_apply) and the assembled CSR (nksr_spmv_csr), bound 3e-6 * (|A| |x|)_i  (SURVEY.md section 8c asks for fp32 rel-tol 1e-5; measured 5e-8 .. 2.3e-7)."""
    import parity_util as pu
    from nksr_amd import solver
    from oracle import density as odens, hierarchy as ohier, kernel as okern, spec
    rec, fld, xyz, nrm = big
    hp = rec.hparams
    L = hp.tree_depth
    inp = fld._solve_inputs
    dev = xyz.device
    so = odens.scale_for_detail_level(xyz.cpu().numpy(), 1.0, hp.voxel_size)
    assert fld.scale == so, 'detail_level scale differs from the oracle: %r vs %r' % (fld.scale, so)
    xs = inp['pos_xyz'].cpu().numpy()                      # scaled, Morton-sorted cloud (what the solve saw)
    H0, _ = spec.half_index(xs, hp.voxel_size)
    pk0 = spec.morton_key(H0 >> 1, 0)
    assert (np.diff(pk0) >= 0).all()
    # (1) hierarchy: one representative point per occupied finest cell defines the same neighbourhood hierarchy (I_d = I_0 >> d)
    first = np.concatenate([[True], pk0[1:] != pk0[:-1]])
    oh = ohier.Hierarchy(hp.voxel_size, L).build_point_neighborhood(xs[first])
    for d in range(L):
        assert np.array_equal(fld.svh.level(d).keys.cpu().numpy(), oh.levels[d].keys), 'level %d voxel keys' % d
    M = oh.num_unknowns
    assert M == fld.svh.num_unknowns
    # (2) kernel rows at sampled sites, features / interpolator weights as the product holds them
    feats = [f.cpu().numpy() for f in fld._feat]
    interps = []
    for d in range(L):
        m = rec.network.interpolators[d]
        interps.append(okern.Interpolator(*[getattr(m, k).detach().cpu().numpy() for k in ('W1', 'b1', 'W2', 'b2', 'W3', 'b3')]))
    psis = [okern.voxel_psi(feats[d], interps[d]) for d in range(L)]
    for d in range(L):
        np.testing.assert_allclose(fld._psi[d].cpu().numpy(), psis[d], rtol=1e-5, atol=1e-6)
    rs = np.random.RandomState(0)
    nxyz = inp['normal_xyz'].cpu().numpy()
    ps = np.sort(rs.choice(xs.shape[0], 6000, replace=False))
    qs = np.sort(rs.choice(nxyz.shape[0], 3000, replace=False))
    _, ov, _ = okern.kernel_rows(oh, feats, interps, psis, xs[ps], False, False)
    hv, _ = fld.kernel_rows(torch.from_numpy(xs[ps]).to(dev), grad=False)
    pu.check('full_size:position_rows', np.abs(hv.cpu().numpy() - ov).max() / np.abs(ov).max(), 3e-6)      # measured 2.1e-7 (round 3)
    _, _, od = okern.kernel_rows(oh, feats, interps, psis, nxyz[qs], True, False)
    _, hd = fld.kernel_rows(torch.from_numpy(nxyz[qs]).to(dev), grad=True, values=False)
    pu.check('full_size:gradient_rows', np.abs(hd.cpu().numpy() - od).max() / np.abs(od).max(), 3e-6)      # measured 1.1e-7
    # (3) operator rows on sampled unknowns
    nk0 = fld.svh.level(0).keys.cpu().numpy()              # normal sites = finest voxel centres (adaptive_depth 1), sorted
    assert hp.adaptive_depth == 1 and nxyz.shape[0] == nk0.shape[0]
    sample = {0: 800, 1: 200, 2: 8, 3: 3}
    U, psel, qsel = [], [], []
    for d, cnt in sample.items():
        lv = oh.levels[d]
        j = np.sort(rs.choice(lv.n, min(cnt, lv.n), replace=False))
        U.append(j + oh.offsets[d])
        cells = (lv.ijk[j][:, None, :] + spec.NBR_OFFSETS[None]).reshape(-1, 3)
        ck = np.unique(spec.morton_key(cells, d))
        for keys_d, out in ((pk0 >> (3 * d), psel), (nk0 >> (3 * d), qsel)):
            out.append(_ranges(np.searchsorted(keys_d, ck, 'left'), np.searchsorted(keys_d, ck, 'right')))
    U = np.concatenate(U)
    psel, qsel = np.unique(np.concatenate(psel)), np.unique(np.concatenate(qsel))
    gc, gv, _ = okern.kernel_rows(oh, feats, interps, psis, xs[psel], False, False)
    qc, _, qd = okern.kernel_rows(oh, feats, interps, psis, nxyz[qsel], True, False)
    import scipy.sparse as sp
    G = okern.rows_to_csr(gc, gv, M).astype(np.float64)
    Q = sp.vstack([okern.rows_to_csr(qc, qd[:, a], M) for a in range(3)]).tocsr().astype(np.float64)
    wp, wn, reg = float(inp['pos_weight']), float(inp['normal_weight']), float(inp['reg_weight'])
    x = rs.randn(M).astype(np.float32)
    x64 = x.astype(np.float64)
    y = wp * (G.T @ (G @ x64)) + wn * (Q.T @ (Q @ x64)) + reg * x64
    mag = wp * (abs(G).T @ (abs(G) @ np.abs(x64))) + wn * (abs(Q).T @ (abs(Q) @ np.abs(x64))) + reg * np.abs(x64)
    xt = torch.from_numpy(x).to(dev)
    op = fld.fused_operator(inp['pos_xyz'], inp['normal_xyz'], inp['normal_value'], inp['pos_weight'], inp['normal_weight'],
                            inp['pos_sorted_keys'], inp['normal_sorted_keys'])
    yf = fld.fused_apply(op, xt, reg).cpu().numpy().astype(np.float64)
    rowptr, cols_p, vals_p, _ = fld.matrix
    yc = solver.spmv(rowptr, cols_p, vals_p, xt).cpu().numpy().astype(np.float64)
    pu.report('full_size:operator_rows', unknowns=int(U.size), pos_sites=int(psel.size), normal_sites=int(qsel.size))
    pu.check('full_size:fused_apply_rows', (np.abs(yf[U] - y[U]) / mag[U]).max(), 3e-6)       # measured 5.2e-8
    pu.check('full_size:spmv_csr_rows', (np.abs(yc[U] - y[U]) / mag[U]).max(), 3e-6)         # measured 2.3e-7
    # the right-hand side rows of the same unknowns: b = w_n Q^T n
    tgt = inp['normal_value'].cpu().numpy()[qsel]
    b = wn * (Q.T @ np.concatenate([tgt[:, a] for a in range(3)]).astype(np.float64))
    bmag = wn * (abs(Q).T @ np.abs(np.concatenate([tgt[:, a] for a in range(3)])).astype(np.float64))
    pu.check('full_size:rhs_rows', (np.abs(fld.rhs.cpu().numpy()[U] - b[U]) / np.maximum(bmag[U], 1e-30)).max(), 3e-6)       # measured 2.3e-7


def test_hierarchy_is_sorted_and_nested(big):
    _, fld, _, _ = big
    svh = fld.svh
    for d in range(svh.depth):
        k = svh.level(d).keys
        assert bool((k[1:] > k[:-1]).all())                       # strictly ascending Morton order, no duplicates
        if d + 1 < svh.depth:                                     # every voxel has its parent one level up
            assert bool((svh.level(d + 1).hash.query((k >> 3).contiguous()) >= 0).all())
        nb = svh.level(d).nbr
        assert bool((nb[:, 13] == torch.arange(nb.shape[0], device=nb.device, dtype=nb.dtype)).all())   # self slot
        # neighbour symmetry on a sample: nbr[nbr[i][s]][26 - s] == i
        idx = torch.randint(0, nb.shape[0], (20000,), device=nb.device)
        for s in (0, 4, 12, 22, 26):
            j = nb[idx, s].long()
            ok = j >= 0
            assert bool((nb[j[ok], 26 - s].long() == idx[ok]).all())


def test_hierarchies_do_not_depend_on_the_lds_dedup_of_the_key_streams(big, monkeypatch):
    """nksr_footprint_keys_dedup drops duplicates before the sort: same SET of keys, hence bit-identical levels
    (both hierarchies of the hot path: point splatting and cell neighbourhoods)."""
    from nksr_amd import svh as svh_mod
    from nksr_amd.nn.network import sort_cloud
    from nksr_amd.svh import SparseFeatureHierarchy, inv_w0_f32
    rec, _, xyz, _ = big
    hp = rec.hparams
    ks, xs, _ = sort_cloud(xyz.contiguous(), xyz.contiguous(), inv_w0_f32(hp.voxel_size))
    built = {}
    for name, limit in (('dedup', 1), ('plain', 1 << 62)):
        monkeypatch.setattr(svh_mod, '_DEDUP_MIN', limit)
        a = SparseFeatureHierarchy(hp.voxel_size, hp.tree_depth, xyz.device).build_point_splatting_sorted(xs, ks)
        b = SparseFeatureHierarchy(hp.voxel_size, hp.tree_depth, xyz.device).build_point_neighborhood_sorted(ks)
        built[name] = [h.level(d).keys for h in (a, b) for d in range(hp.tree_depth)]
    for u, v in zip(built['dedup'], built['plain']):
        assert u.numel() > 0 and torch.equal(u, v)
    ref = SparseFeatureHierarchy(hp.voxel_size, hp.tree_depth, xyz.device).build_point_splatting(xs)     # per-point keys, no shortcuts
    for d in range(hp.tree_depth):
        assert torch.equal(built['dedup'][d], ref.level(d).keys)


def test_matrix_is_exactly_symmetric_and_positive(big):
    from nksr_amd import solver
    _, fld, _, _ = big
    rowptr, cols_p, vals_p, diag = fld.matrix
    M = rowptr.numel() - 1
    nnz = int(rowptr[-1])
    assert nnz > 3e8 and M > 1e6
    cols, vals = solver.csr_logical(rowptr, cols_p, vals_p)
    g = torch.Generator(device='cpu').manual_seed(0)
    rp = rowptr.long()
    rows = torch.repeat_interleave(torch.arange(M, device=rowptr.device), rp[1:] - rp[:-1])
    assert bool((cols[rp[1:] - 1].long() == torch.arange(M, device=rowptr.device)).all())      # diagonal closes every row
    assert bool((vals[rp[1:] - 1] == diag).all())
    # all 3.8e8 entries: the multiset {(i, j, a_ij)} equals {(j, i, a_ij)} bit for bit
    a, ia = torch.sort(rows * M + cols.long())
    assert bool((a[1:] != a[:-1]).all()), 'duplicate (row, col) entries'
    b, ib = torch.sort(cols.long() * M + rows)
    assert bool((a == b).all()), 'structural asymmetry'
    assert bool((vals[ia] == vals[ib]).all()), 'mirrored value differs'
    del a, b, ia, ib, rows
    # bilinear-form symmetry and positivity through the product SpMV (fp64 accumulation of the dots)
    x = torch.randn(M, generator=g).to(rowptr.device)
    y = torch.randn(M, generator=g).to(rowptr.device)
    Ax, Ay = solver.spmv(rowptr, cols_p, vals_p, x), solver.spmv(rowptr, cols_p, vals_p, y)
    a, b = float((y.double() * Ax.double()).sum()), float((x.double() * Ay.double()).sum())
    assert abs(a - b) <= 1e-5 * max(abs(a), abs(b), float(Ax.double().norm() * y.double().norm()) * 1e-2)
    assert float((x.double() * Ax.double()).sum()) > 0
    assert bool((diag > 0).all())


def test_solution_satisfies_the_system(big):
    from nksr_amd import solver
    _, fld, _, _ = big
    rowptr, cols_p, vals_p, _ = fld.matrix
    res = fld.rhs.double() - solver.spmv(rowptr, cols_p, vals_p, fld.alpha).double()
    rel = float(res.norm() / fld.rhs.double().norm())
    assert rel <= 2e-5, rel                                       # solver_tol 1e-5, independent residual
    assert fld.solve_info['rel_residual'] <= 1e-5 and fld.solve_info['iters'] < 200


def test_matrix_free_solve_agrees_at_full_size(big):
    """fused_mode=True on the 1M-point cloud: the operator equals the assembled matrix on a random vector, the solve
    reaches the tolerance against the ASSEMBLED system (independent residual) and defines the same field."""
    from nksr_amd import solver
    rec, fld, xyz, nrm = big
    ff = rec.reconstruct(xyz, nrm, detail_level=1.0, fused_mode=True)
    assert ff.solve_info['fused'] and ff.solve_info['rel_residual'] <= 1e-5 and ff.matrix is None
    rowptr, cols_p, vals_p, diag = fld.matrix
    assert float(((ff.diag - diag).abs() / diag).max()) <= 1e-5
    assert float((ff.rhs - fld.rhs).abs().max()) <= 1e-5 * float(fld.rhs.abs().max())
    res = fld.rhs.double() - solver.spmv(rowptr, cols_p, vals_p, ff.alpha).double()
    assert float(res.norm() / fld.rhs.double().norm()) <= 3e-5
    sel = torch.randperm(xyz.shape[0], device=xyz.device)[:100000]
    q = (xyz[sel] + 0.02).contiguous()
    fa, fb = fld.evaluate_f(q).value, ff.evaluate_f(q).value
    assert float((fa - fb).abs().max()) <= 1e-3 * float(fa.abs().max())          # both solved to 1e-5


def test_field_fits_the_input(big):
    _, fld, xyz, nrm = big
    sel = torch.randperm(xyz.shape[0], device=xyz.device)[:200000]
    res = fld.evaluate_f(xyz[sel].contiguous(), grad=True)
    f, g = res.value, res.gradient
    w = 0.1 / fld.scale                                           # finest voxel in world units
    assert float(f.abs().median()) < 0.15                         # noise sigma = 0.01 ~ 0.1 voxel
    cosang = -(g * nrm[sel]).sum(1) / g.norm(dim=1).clamp_min(1e-12)
    assert float(cosang.mean()) > 0.95 and float((cosang > 0).float().mean()) > 0.995
    # f changes sign across the surface (f > 0 inside, models/loss.py:192-196): step one voxel along the normal
    fin = fld.evaluate_f((xyz[sel] - w * nrm[sel]).contiguous()).value
    fout = fld.evaluate_f((xyz[sel] + w * nrm[sel]).contiguous()).value
    assert float((fin > 0).float().mean()) > 0.97 and float((fout < 0).float().mean()) > 0.97


@pytest.mark.parametrize('mise_iter', [0, 1])
def test_mesh_is_watertight_and_on_the_data(big, mise_iter):
    _, fld, xyz, _ = big
    mesh = fld.extract_dual_mesh(mise_iter=mise_iter)
    f = mesh.f.long()
    V = mesh.v.shape[0]
    assert f.shape[0] > 5e5 and int(f.min()) == 0 and int(f.max()) == V - 1
    e = torch.cat([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
    key = torch.minimum(e[:, 0], e[:, 1]) * V + torch.maximum(e[:, 0], e[:, 1])
    _, cnt = torch.unique(key, return_counts=True)
    # manifold everywhere; open edges only where the noisy level set leaves the band of active voxels
    assert int(cnt.max()) == 2, 'non-manifold edges: %s' % torch.bincount(cnt)[:6].tolist()
    assert int((cnt == 1).sum()) <= 2e-4 * cnt.numel(), 'open edges: %s' % torch.bincount(cnt)[:6].tolist()
    # orientation consistency: every directed edge appears exactly once
    dkey = e[:, 0] * V + e[:, 1]
    assert torch.unique(dkey).numel() == dkey.numel()
    # vertices sit on the zero level set and next to the data
    fv = fld.evaluate_f(mesh.v).value
    assert float(fv.abs().max()) < 0.5
    assert bool(torch.isfinite(mesh.v).all())


@pytest.mark.parametrize('mise_iter', [0, 1])
def test_full_size_mesh_window_matches_the_oracle_mesher(big, mise_iter):
    """The 1 M-point mesh against oracle/meshing.py on a window of ~20^3 finest voxels (the oracle cannot mesh 750 000 voxels, it can
    mesh a few thousand): the oracle mesher runs on the window's voxels with the HIP field's lattice values; triangles of the
    window's inner cells index-exact and in the same order, vertices within 1e-4 voxel."""
    import parity_util as pu
    rec, fld, xyz, nrm = big
    g0 = fld.svh.level(0)
    centre = g0.ijk[g0.num_voxels // 2].tolist()
    pu.window_mesh_check('full_size:mesh_window[mise=%d]' % mise_iter, fld, fld.scale, mise_iter, centre, 10, w0=rec.hparams.voxel_size)


# ---- BASELINE.json configs[4] at its full size: the 64-chunk batch of bench.py -------------------------------------------------------
def _bench():
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec_ = importlib.util.spec_from_file_location('bench_mod_fs', os.path.join(root, 'bench.py'))
    m = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(m)
    return m


def test_bench_scale_batch_is_its_solo_chunks_bit_for_bit_and_its_operator_rows_match_the_oracle():
    """The scene bench.py times (10 M points, 8 x 8 tiles, tree_depth 5, all 64 chunks of the rank as ONE block-diagonal solve):
      * every chunk solved ALONE (chunk_batch_points = 1: 64 batches of one) gives the voxel keys, the iteration count and the
        coefficients of its segment of the 64-chunk batch BIT FOR BIT -- the reference solves its chunks one after the other
        (examples/recons_by_chunk.py:26-29); batching them must not change a single bit;
      * one chunk of the batch against the oracle: its voxel keys at all five levels exactly, and rows of y = A x of ~1 000 sampled
        unknowns of that chunk -- x random over the WHOLE batch, so a leak between the diagonal blocks would show -- against the
        oracle evaluating all constraint rows in their support, bound 3e-6 (|A| |x|)_i as at the configs[2] size."""
    import parity_util as pu
    import scipy.sparse as sp
    import nksr_amd
    from nksr_amd import configs
    from oracle import hierarchy as ohier, kernel as okern, spec
    b = _bench()
    dev = torch.device('cuda:0')
    rec = nksr_amd.Reconstructor(dev, hparams=configs.get_hparams('ks', tree_depth=5))
    rec.keep_solve_inputs = True
    hp = rec.hparams
    L = hp.tree_depth
    xyz, nrm, scale, owner, bounds, n_scene, _ = b.terrain_setup(rec, dev, 10_000_000, 0, 1)
    assert n_scene == 10_000_000
    kw = dict(detail_level=None, chunk_size=b.TILE * scale, sharded_input=True, chunk_owner=owner, chunk_bounds=bounds)
    fld = rec.reconstruct(xyz, nrm, **kw)
    assert len(fld.parts) == 1 and len(fld.parts[0].ids) == 64
    part = fld.parts[0]
    bf = part.field
    inp = bf._solve_inputs
    seg = inp['segments']
    info = bf.solve_info['segment_info'].cpu().numpy()
    lo, hi = seg.lo.cpu().numpy(), seg.hi.cpu().numpy()            # [64, L] unknown ranges of every chunk
    off = bf.svh.offsets
    batch = {}
    for i, c in enumerate(part.ids):
        batch[c] = ([bf.svh.level(d).keys[lo[i, d] - off[d]:hi[i, d] - off[d]].clone() for d in range(L)],
                    torch.cat([bf.alpha[lo[i, d]:hi[i, d]] for d in range(L)]).clone(), int(info[i, 0]))
    assert max(v[2] for v in batch.values()) <= 40 and float(info[:, 1].max()) <= 1e-5
    assert bf.solve_info['jacobi_fallbacks'] == 0          # no segment lost its coarse-level block (a fallback changes iteration counts, not results)

    # ---- one chunk of the batch against the oracle
    i = part.ids.index(27)                                          # chunk id 27 = tile (3, 3): the one the oracle solved in full (tests/golden)
    c = part.ids[i]
    pk = inp['pos_sorted_keys']
    p0, p1 = [int(v) for v in torch.searchsorted(pk, torch.stack([seg.key_lo[i], seg.key_hi[i]])).tolist()]
    xs = inp['pos_xyz'][p0:p1].cpu().numpy()
    H0, _ = spec.half_index(xs, hp.voxel_size)
    pk0 = spec.morton_key(H0 >> 1, 0)
    assert np.array_equal(pk0, pk[p0:p1].cpu().numpy())
    first = np.concatenate([[True], pk0[1:] != pk0[:-1]])
    oh = ohier.Hierarchy(hp.voxel_size, L).build_point_neighborhood(xs[first])
    for d in range(L):
        assert np.array_equal(batch[c][0][d].cpu().numpy(), oh.levels[d].keys), 'chunk %d level %d voxel keys' % (c, d)
    Mc = oh.num_unknowns
    feats = [bf._feat[d][lo[i, d] - off[d]:hi[i, d] - off[d]].cpu().numpy() for d in range(L)]
    interps = []
    for d in range(L):
        m = rec.network.interpolators[d]
        interps.append(okern.Interpolator(*[getattr(m, k).detach().cpu().numpy() for k in ('W1', 'b1', 'W2', 'b2', 'W3', 'b3')]))
    psis = [okern.voxel_psi(feats[d], interps[d]) for d in range(L)]
    assert hp.adaptive_depth == 1
    q0, q1 = int(lo[i, 0]), int(hi[i, 0])                          # normal sites = the chunk's finest voxel centres
    nxyz = inp['normal_xyz'][q0:q1].cpu().numpy()
    nk0 = oh.levels[0].keys
    rs = np.random.RandomState(0)
    U, psel, qsel = [], [], []
    for d, cnt in {0: 700, 1: 200, 2: 60, 3: 10, 4: 3}.items():
        lv = oh.levels[d]
        j = np.sort(rs.choice(lv.n, min(cnt, lv.n), replace=False))
        U.append(j + oh.offsets[d])
        cells = (lv.ijk[j][:, None, :] + spec.NBR_OFFSETS[None]).reshape(-1, 3)
        ck = np.unique(spec.morton_key(cells, d))
        for keys_d, out in ((pk0 >> (3 * d), psel), (nk0 >> (3 * d), qsel)):
            out.append(_ranges(np.searchsorted(keys_d, ck, 'left'), np.searchsorted(keys_d, ck, 'right')))
    U = np.concatenate(U)
    psel, qsel = np.unique(np.concatenate(psel)), np.unique(np.concatenate(qsel))
    gc, gv, _ = okern.kernel_rows(oh, feats, interps, psis, xs[psel], False, False)
    qc, _, qd = okern.kernel_rows(oh, feats, interps, psis, nxyz[qsel], True, False)
    G = okern.rows_to_csr(gc, gv, Mc).astype(np.float64)
    Q = sp.vstack([okern.rows_to_csr(qc, qd[:, a], Mc) for a in range(3)]).tocsr().astype(np.float64)
    # the chunk's own solver weights (models/nksr_net.py:103-111): the batch carries sqrt(weight) per site
    wp, wn = float(inp['pos_weight'][p0]) ** 2, float(inp['normal_weight'][q0]) ** 2
    assert abs(wp - hp.solver.pos_weight / (p1 - p0)) <= 1e-6 * wp and abs(wn - hp.solver.normal_weight / (q1 - q0) * hp.voxel_size ** 2) <= 1e-6 * wn
    M = bf.svh.num_unknowns
    xb = torch.randn(M, device=dev)
    # chunk-local -> batch unknown index (level-major inside the chunk, level-major inside the batch)
    glob = np.concatenate([np.arange(lo[i, d], hi[i, d]) for d in range(L)])
    x64 = xb.cpu().numpy()[glob].astype(np.float64)
    y = wp * (G.T @ (G @ x64)) + wn * (Q.T @ (Q @ x64)) + x64
    mag = wp * (abs(G).T @ (abs(G) @ np.abs(x64))) + wn * (abs(Q).T @ (abs(Q) @ np.abs(x64))) + np.abs(x64)
    op = bf.fused_operator(inp['pos_xyz'], inp['normal_xyz'], inp['normal_value'], inp['pos_weight'], inp['normal_weight'],
                           inp['pos_sorted_keys'], inp['normal_sorted_keys'], segments=seg)
    yf = bf.fused_apply(op, xb, 1.0).cpu().numpy()[glob].astype(np.float64)
    del op
    pu.report('bench_scale:operator_rows', chunk=c, unknowns=int(U.size), pos_sites=int(psel.size), normal_sites=int(qsel.size), batch_M=M)
    pu.check('bench_scale:fused_apply_rows', (np.abs(yf[U] - y[U]) / mag[U]).max(), 3e-6)

    # ---- the mesh at a chunk SEAM against the oracle mesher (a window of ~20^3 finest voxels across the plane between chunk
    # columns 3 | 4, inside the blend zone: both chunks' fields and the partition-of-unity weights enter every lattice value)
    w0 = hp.voxel_size
    seam_x = int(round(4 * b.TILE * scale / w0))
    ug = fld.svh.level(0)
    cand = torch.nonzero(ug.ijk[:, 0] == seam_x).reshape(-1)
    assert cand.numel() > 0
    centre = ug.ijk[cand[cand.numel() // 2]].tolist()
    pu.window_mesh_check('bench_scale:seam_mesh_window', fld, getattr(fld, 'scale', 1.0), 1, centre, 10, w0=w0)

    # ---- ONE FULL CHUNK against the oracle's complete solve of it (tests/golden/scene_chunk27_golden.npz, oracle/make_golden_scene_chunk.py:
    # the reference solves its chunks one at a time, examples/recons_by_chunk.py:26-30 -- the oracle did exactly that for chunk 27)
    gpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'scene_chunk%d_golden.npz' % c)
    assert os.path.exists(gpath), 'missing fixture %s (python -m oracle.make_golden_scene_chunk %d)' % (gpath, c)
    gold = np.load(gpath)
    assert int(gold['points']) == p1 - p0 and float(gold['scale']) == float(scale)
    assert [int(v) for v in gold['level_n']] == [int(hi[i, d] - lo[i, d]) for d in range(L)]
    for d in range(L):
        assert hashlib.sha256(np.ascontiguousarray(batch[c][0][d].cpu().numpy().astype(np.int64)).tobytes()).hexdigest() == str(gold['level_key_sha256'][d]), \
            'chunk %d level %d: voxel keys differ from the oracle run' % (c, d)
    amax = float(gold['alpha_absmax'])
    pq = torch.from_numpy(gold['probe_xyz']).to(dev)
    ev = bf._evaluate_f_model(pq, True)
    fscale = float(np.abs(gold['probe_f']).max())
    pu.report('bench_scale:golden_chunk', chunk=c, M=int(gold['alpha'].size), oracle_iters=int(gold['iters']), hip_iters=batch[c][2],
              alpha_rel_at_default_tol=float(np.abs(batch[c][1].cpu().numpy() - gold['alpha']).max() / amax))
    pu.check('bench_scale:golden_chunk:field_rel', float(np.abs(ev.value.cpu().numpy() - gold['probe_f']).max() / fscale), 1e-4)
    pu.check('bench_scale:golden_chunk:gradient_abs', float(np.abs(ev.gradient.cpu().numpy() - gold['probe_grad']).max()), 1e-3)

    # ---- every chunk alone
    del fld, part, bf, inp, seg, ev
    torch.cuda.empty_cache()
    rec.keep_solve_inputs = False
    rec.chunk_batch_points = 1
    solo = rec.reconstruct(xyz, nrm, **kw)
    assert len(solo.parts) == 64 and all(len(p.ids) == 1 for p in solo.parts)
    for p in solo.parts:
        c = p.ids[0]
        keys_b, alpha_b, it_b = batch[c]
        for d in range(L):
            assert torch.equal(p.field.svh.level(d).keys, keys_b[d]), 'chunk %d level %d: voxel keys differ between batch and solo' % (c, d)
        assert p.field.solve_info['iters'] == it_b, 'chunk %d: %d iterations alone, %d in the batch' % (c, p.field.solve_info['iters'], it_b)
        assert torch.equal(p.field.alpha, alpha_b), 'chunk %d: coefficients differ between batch and solo' % c
    # ---- converged coefficients: the same scene solved to 1e-6, chunk 27's segment against the oracle's alpha (1e-4)
    del solo
    torch.cuda.empty_cache()
    rec.chunk_batch_points = None
    rec.keep_solve_inputs = True
    tight = rec.reconstruct(xyz, nrm, solver_tol=1e-6, **kw)
    tb = tight.parts[0].field
    tseg = tb._solve_inputs['segments']
    tlo, thi = tseg.lo.cpu().numpy(), tseg.hi.cpu().numpy()
    ti = tight.parts[0].ids.index(27)
    a27 = torch.cat([tb.alpha[tlo[ti, d]:thi[ti, d]] for d in range(L)]).cpu().numpy()
    pu.check('bench_scale:golden_chunk:alpha_rel', float(np.abs(a27 - gold['alpha']).max() / float(gold['alpha_absmax'])), pu.ALPHA_TOL)
