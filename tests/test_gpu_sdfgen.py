"""ext.sdfgen.sdf_from_points (training ground truth; reference ext/sdfgen/sdf_from_points.cu, call sites models/loss.py:85,
dataset/av_gt_geometry.py:64-76) against the oracle restatement of its source (oracle/sdfgen.py, exact kNN through scipy) -- and
both against the REFERENCE'S OWN extension, compiled from its sources for gfx950 (oracle/_ref/nksr_sdfgen.so, recipe
oracle/build_ref.py): the one piece of this project whose parity is pinned on reference code that runs."""
import os

import numpy as np
import pytest
import torch

import parity_util as pu

pytestmark = pytest.mark.gpu


def _cloud(n=20000, seed=0, noise=0.002):
    rs = np.random.RandomState(seed)
    v = rs.randn(n, 3)
    nrm = (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)
    xyz = (nrm * 0.5 + rs.randn(n, 3) * noise).astype(np.float32)
    return xyz, nrm


def _queries(xyz, nrm, m=6000, seed=1, eps=0.03):
    rs = np.random.RandomState(seed)
    band = xyz[rs.randint(0, len(xyz), m)] + nrm[rs.randint(0, len(xyz), m)] * rs.randn(m, 1).astype(np.float32) * eps
    far = rs.uniform(-1.5, 1.5, (m // 10, 3)).astype(np.float32)          # far from every reference point: the coarse-grid retry
    return np.concatenate([band, far]).astype(np.float32)


@pytest.mark.parametrize('case', ['loss', 'dataset', 'imls'])
def test_sdf_from_points_matches_the_oracle(case):
    from oracle import sdfgen as osdf
    import ext                      # the reference's import name
    dev = torch.device('cuda:0')
    xyz, nrm = _cloud()
    q = _queries(xyz, nrm)
    kw = {'loss': dict(nb_points=8, stdv=0.02), 'dataset': dict(nb_points=8, stdv=3.0, adaptive_knn=8),
          'imls': dict(nb_points=8, stdv=0.05, imls=True)}[case]
    t = lambda a: torch.from_numpy(a).to(dev)
    out = ext.sdfgen.sdf_from_points(t(q), t(xyz), t(nrm), compute_grad=True, **kw)
    assert len(out) == 2 and out[0].shape == (len(q),) and out[1].shape == (len(q), 3) and out[0].dtype == torch.float32
    only = ext.sdfgen.sdf_from_points(t(q), t(xyz), t(nrm), **kw)
    assert len(only) == 1 and torch.equal(only[0], out[0])
    ref_s, ref_g = osdf.sdf_from_points(q, xyz, nrm, compute_grad=True, **kw)
    s, g = out[0].cpu().numpy(), out[1].cpu().numpy()
    # sign votes / nearest-neighbour switches flip where a query is equidistant to two neighbours within fp32 rounding: compare
    # where the oracle's decision has margin, count the rest
    diff = np.abs(s - ref_s)
    bad = diff > 1e-5 + 1e-5 * np.abs(ref_s)
    pu.report('sdfgen[%s]' % case, max_abs_err=float(diff[~bad].max()), flipped=int(bad.sum()), queries=len(q))
    assert bad.mean() <= 2e-3, int(bad.sum())
    gd = np.abs(g - ref_g).max(1)
    assert (gd[~bad] <= 1e-4).mean() >= 0.998
    # the negated value is the training target (models/loss.py:85): inside the sphere positive, outside negative
    inside = np.linalg.norm(q, axis=1) < 0.45
    outside = np.linalg.norm(q, axis=1) > 0.55
    assert (-s[inside] > 0).mean() > 0.99 and (-s[outside] < 0).mean() > 0.99


@pytest.mark.parametrize('case', ['loss', 'dataset', 'imls'])
def test_sdf_from_points_matches_the_reference_binary(case):
    """The reference's kernels (kd-tree kNN of ext/common/kdtree_cuda.cu + the estimator of ext/sdfgen/sdf_from_points.cu:32-140)
    run on this GPU through oracle/_ref/nksr_sdfgen.so: nksr_amd's single-kernel implementation and the numpy restatement must
    both reproduce its values and gradients -- the same fp32 formula over the same k nearest neighbours, so the agreement is at
    rounding level except where two neighbours are equidistant within fp32 (counted, bounded)."""
    from oracle import build_ref, sdfgen as osdf
    import ext
    ref = build_ref.load()
    if ref is None and os.path.isdir(build_ref.REF):      # (dev container: the fixture builds the checker itself; the GPU box gets it with the snapshot)
        build_ref.build(verbose=False)
        ref = build_ref.load()
    if ref is None:
        pytest.skip('oracle/_ref/nksr_sdfgen.so is missing: __graft_entry__.build() / `python -m oracle.build_ref` makes it where /root/reference exists (it travels with the snapshot)')
    dev = torch.device('cuda:0')
    xyz, nrm = _cloud()
    q = _queries(xyz, nrm)
    kw = {'loss': dict(nb_points=8, stdv=0.02), 'dataset': dict(nb_points=8, stdv=3.0, adaptive_knn=8),
          'imls': dict(nb_points=8, stdv=0.05, imls=True)}[case]
    t = lambda a: torch.from_numpy(a).to(dev)
    r = ref.sdf_from_points(t(q), t(xyz), t(nrm), kw['nb_points'], kw['stdv'], True, kw.get('imls', False), kw.get('adaptive_knn', 0))
    torch.cuda.synchronize()
    rs, rg = r[0].cpu().numpy(), r[1].cpu().numpy()
    assert rs.shape == (len(q),) and rg.shape == (len(q), 3) and np.isfinite(rs).all()
    out = ext.sdfgen.sdf_from_points(t(q), t(xyz), t(nrm), compute_grad=True, **kw)
    os_, og = osdf.sdf_from_points(q, xyz, nrm, compute_grad=True, **kw)
    for name, s, g in (('hip', out[0].cpu().numpy(), out[1].cpu().numpy()), ('oracle', os_, og)):
        diff = np.abs(s - rs)
        bad = diff > 1e-5 + 1e-5 * np.abs(rs)
        gd = np.abs(g - rg).max(1)
        tie = gd > 1e-4                   # the nearest neighbour is a tie within fp32 (two reference points at the same distance): either is right
        pu.report('sdfgen_vs_reference_binary[%s,%s]' % (case, name), max_abs_err=float(diff[~bad].max()), flipped=int(bad.sum()),
                  grad_err=float(gd[~bad & ~tie].max()), nearest_neighbour_ties=int((tie & ~bad).sum()), queries=len(q))
        assert bad.mean() <= 2e-3, (name, int(bad.sum()))
        assert (gd[~bad] <= 1e-4).mean() >= 0.998, name


@pytest.mark.parametrize('k', [8, 20])
def test_octree_search_equals_the_single_grid_rounds_and_hands_on_what_it_cannot_reach(k, monkeypatch):
    """The default search (every scale in one launch, csrc/knn.hip k_sdf_pyramid) against the single-grid rounds it replaced and
    against exact kNN -- with queries at every distance: in the band, across the bounding box, and 100 cloud diameters away
    (beyond the coarsest level: those come back unanswered and take the rounds)."""
    from oracle import sdfgen as osdf
    import ext
    dev = torch.device('cuda:0')
    xyz, nrm = _cloud(30000, seed=3)
    rs = np.random.RandomState(5)
    q = np.concatenate([_queries(xyz, nrm, 20000, seed=4), rs.uniform(-6, 6, (500, 3)), rs.uniform(-100, 100, (60, 3))]).astype(np.float32)
    t = lambda a: torch.from_numpy(a).to(dev)
    kw = dict(nb_points=k, stdv=0.05, compute_grad=True)
    for imls in (False, True):
        monkeypatch.setenv('NKSR_SDFGEN_SEARCH', 'rounds')
        a = ext.sdfgen.sdf_from_points(t(q), t(xyz), t(nrm), imls=imls, **kw)
        monkeypatch.delenv('NKSR_SDFGEN_SEARCH')
        b = ext.sdfgen.sdf_from_points(t(q), t(xyz), t(nrm), imls=imls, **kw)
        ref_s, _ = osdf.sdf_from_points(q, xyz, nrm, k, 0.05, compute_grad=True, imls=imls)
        sa, sb = a[0].cpu().numpy(), b[0].cpu().numpy()
        tol = 1e-5 + 1e-5 * np.abs(ref_s)
        pu.report('sdfgen octree[k=%d,%s]' % (k, 'imls' if imls else 'vote'), vs_rounds=float(np.abs(sa - sb).max()),
                  differ_from_exact_knn=int((np.abs(sb - ref_s) > tol).sum()), queries=len(q))
        assert (np.abs(sa - sb) > tol).mean() <= 1e-3
        assert (np.abs(sb - ref_s) > tol).mean() <= 2e-3
        assert ((a[1] - b[1]).abs().amax(1).cpu().numpy() > 1e-4).mean() <= 2e-3


def test_sdf_from_points_argument_errors():
    import ext
    dev = torch.device('cuda:0')
    xyz, nrm = _cloud(100)
    t = lambda a: torch.from_numpy(a).to(dev)
    with pytest.raises(RuntimeError):
        ext.sdfgen.sdf_from_points(t(xyz[:5]), t(xyz[:4]), t(nrm[:4]), 8, 0.02)          # fewer reference points than nb_points
    with pytest.raises(RuntimeError):
        ext.sdfgen.sdf_from_points(torch.from_numpy(xyz[:5]), t(xyz), t(nrm), 8, 0.02)   # CPU queries
    with pytest.raises(RuntimeError):
        ext.sdfgen.sdf_from_points(t(xyz[:5]), t(xyz), t(nrm), 8, 0.0)                   # stdv must be > 0
