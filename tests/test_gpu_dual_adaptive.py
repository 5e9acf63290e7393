"""field.extract_dual_mesh on the ADAPTIVE dual graph (field.dual_graph = 'adaptive': cells as large as the hierarchy level that
carries them -- LayerField(dec_svh, adaptive_depth), reference models/nksr_net.py:132,214,284) against its specification
oracle/dual_adaptive.py, fed with the SAME field values (the HIP field evaluated at the oracle's sample positions), so what is compared
is the mesher: leaves, dual cells, MISE splits, table look-up, vertex naming -- triangles index for index, vertices to rounding."""
import numpy as np
import pytest
import torch

import parity_util as pu

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device('cuda:0')


@pytest.fixture(scope='module')
def sphere_field():
    import nksr_amd
    from nksr_amd import configs
    n = 4000
    k = np.arange(n) + 0.5
    phi, z = np.pi * (1 + 5 ** 0.5) * k, 1 - 2 * k / n
    nrm = np.stack([np.cos(phi) * np.sqrt(1 - z * z), np.sin(phi) * np.sqrt(1 - z * z), z], 1).astype(np.float32)
    xyz = (nrm * np.float32(0.45) + np.float32([0.003, -0.002, 0.001])).astype(np.float32)
    rec = nksr_amd.Reconstructor(_dev(), hparams=configs.get_hparams('ks', adaptive_depth=2))
    fld = rec.reconstruct(torch.from_numpy(xyz).to(_dev()), torch.from_numpy(nrm).to(_dev()), voxel_size=0.04, solver_tol=1e-6)
    assert fld.meshing_depth == 2
    return fld


def _eval(fld):
    def ev(p):
        t = torch.from_numpy(np.ascontiguousarray(p, np.float32)).to(_dev())
        return fld._evaluate_f_model(t, False, max_points=1 << 22).value.cpu().numpy()
    return ev


def _mask(fld):
    def mk(p):
        m = fld.mask_vertices(torch.from_numpy(np.ascontiguousarray(p, np.float32)).to(_dev()))
        return np.ones(len(p), bool) if m is None else m.cpu().numpy()
    return mk


def _compare(name, mesh, ov, of, scale):
    gv, gf = mesh.v.cpu().numpy() * np.float32(scale), mesh.f.cpu().numpy()
    pu.report(name, triangles=len(gf), oracle_triangles=len(of), vertices=len(gv))
    assert gf.shape == of.shape and np.array_equal(gf, of), '%s: triangles differ' % name
    pu.check(name + ':vertices', float(np.abs(gv - ov).max()) if len(ov) else 0.0, 2e-6)


def _block(lo, hi):
    r = np.arange(lo, hi)
    return np.stack(np.meshgrid(r, r, r, indexing='ij'), -1).reshape(-1, 3)


CORNERS = np.array([[a, b, c] for a in (0, 1) for b in (0, 1) for c in (0, 1)])


def _pattern(kind, seed=0):
    rs = np.random.RandomState(seed)
    l1 = _block(-8, 8)                                     # level-1 voxels (0.2 model units): [-1.6, 1.6]^3 around the sphere of radius 1.125
    refine = {'half': l1[:, 0] < 0, 'random': rs.rand(len(l1)) < 0.4, 'none': np.zeros(len(l1), bool)}[kind]
    ch = (l1[refine][:, None, :] * 2 + CORNERS[None]).reshape(-1, 3)
    if kind == 'random':
        ch = ch[rs.rand(len(ch)) < 0.8]                     # partial octants: virtual children
    return [ch, l1]


@pytest.mark.parametrize('mise_iter,upsample', [(0, 1), (1, 1), (2, 1), (0, 2)])
def test_adaptive_dual_graph_of_the_fields_own_hierarchy_matches_the_specification(sphere_field, mise_iter, upsample):
    from oracle import dual_adaptive as da
    fld = sphere_field
    levels = [fld.svh.level(d).ijk.cpu().numpy() for d in range(fld.meshing_depth)]
    ov, of = da.extract(fld.svh.voxel_size, levels, _eval(fld), mise_iter, upsample, mask_fn=_mask(fld))
    fld.dual_graph = 'adaptive'
    try:
        mesh = fld.extract_dual_mesh(mise_iter=mise_iter, grid_upsample=upsample)
    finally:
        fld.dual_graph = 'lattice'
    assert len(of) > 1000
    _compare('dual_adaptive[own,mise=%d,U=%d]' % (mise_iter, upsample), mesh, ov, of, fld.scale)


@pytest.mark.parametrize('kind', ['half', 'random', 'none'])
@pytest.mark.parametrize('mise_iter,upsample', [(0, 1), (1, 1), (1, 2)])
def test_adaptive_dual_graph_on_mixed_level_patterns_matches_the_specification(sphere_field, kind, mise_iter, upsample):
    """Octrees the seeded structure head does not produce -- half the block refined, a random 40 % with partial octants, nothing
    refined -- around the same field: level transitions, virtual children, degenerate hexahedra."""
    from nksr_amd import meshing
    from oracle import dual_adaptive as da
    fld = sphere_field
    levels = _pattern(kind)
    info = {}
    ov, of = da.extract(fld.svh.voxel_size, levels, _eval(fld), mise_iter, upsample, mask_fn=_mask(fld), info=info)
    mesh = meshing._extract_adaptive(fld, mise_iter, upsample, -1, level_ijk=levels)
    assert len(of) > 500
    _compare('dual_adaptive[%s,mise=%d,U=%d]' % (kind, mise_iter, upsample), mesh, ov, of, fld.scale)
    # the cell table itself: sizes, keys and sampled values of the last MISE level
    last = info['levels'][-1]
    assert np.array_equal(mesh.cell_lam.cpu().numpy(), last['lam'])
    assert np.array_equal(mesh.cell_f.cpu().numpy(), last['f'])


def test_uniform_hierarchy_gives_the_lattice_mesh(sphere_field):
    """One level, no upsampling, no MISE: the adaptive dual graph IS the lattice of level-0 centres -- same triangles, same
    vertices as the default mesher."""
    fld = sphere_field
    depth = fld.meshing_depth
    fld.meshing_depth = 1
    try:
        a = fld.extract_dual_mesh(mise_iter=0)
        fld.dual_graph = 'adaptive'
        b = fld.extract_dual_mesh(mise_iter=0)
    finally:
        fld.dual_graph, fld.meshing_depth = 'lattice', depth
    assert a.f.shape[0] > 1000 and torch.equal(a.f, b.f) and torch.equal(a.v, b.v)


@pytest.mark.parametrize('mise_iter', [0, 1])
def test_chunked_field_meshes_on_the_adaptive_dual_graph_of_its_union_hierarchy(mise_iter):
    """reconstruct(chunk_size=) with adaptive_depth 2, held by one process: ``dual_graph='adaptive'`` meshes the union hierarchy of the
    chunks (levels 0 and 1 on the global lattice) with the BLENDED field -- against oracle/dual_adaptive.py on the same two levels,
    fed with the same blended values (the seam crosses the window: both chunks and the partition-of-unity weights enter)."""
    import nksr_amd
    from nksr_amd import configs, utils
    from oracle import dual_adaptive as da
    xyz, nrm = utils.synth_terrain_patch(16000, seed=7, extent=(8.0, 4.0))
    rec = nksr_amd.Reconstructor(_dev(), hparams=configs.get_hparams('ks', adaptive_depth=2))
    rec.dual_graph = 'adaptive'
    fld = rec.reconstruct(torch.from_numpy(xyz).to(_dev()), torch.from_numpy(nrm).to(_dev()), detail_level=None, chunk_size=4.0 + 1e-3)
    assert len(fld.fields) >= 2 and fld.meshing_depth == 2 and fld.dual_graph == 'adaptive'
    levels = [fld.svh.level(d).ijk.cpu().numpy() for d in range(2)]
    assert all(len(l) for l in levels)
    ov, of = da.extract(fld.svh.voxel_size, levels, _eval(fld), mise_iter, 1, mask_fn=_mask(fld))
    mesh = fld.extract_dual_mesh(mise_iter=mise_iter)
    assert len(of) > 1000
    _compare('dual_adaptive[chunked,mise=%d]' % mise_iter, mesh, ov, of, getattr(fld, 'scale', 1.0))
