"""CPU self-test of the topology comparison used by the GPU parity tests (tests/parity_util.py): the oracle
mesh against the oracle mesh of a slightly perturbed field -- identical when the perturbation is zero, and
every differing triangle inside the near-threshold ("tainted") cells when it is not."""
import numpy as np

from conftest import make_cloud
import parity_util as pu


def _field():
    from oracle import pipeline
    xyz, nrm = make_cloud('sphere', 1200, 0.005, 0)
    xs = (xyz * np.float32(1.5)).astype(np.float32)
    return pipeline.reconstruct(xs, nrm, tol=1e-6), xs


def test_compare_meshes_localises_sign_flips():
    from oracle import meshing, pipeline
    fld, xs = _field()
    ev = lambda p: pipeline.evaluate(fld, p)[0]
    info = {}
    ov, of = meshing.extract(fld['voxel_size'], fld['hier'].levels[0], ev, 1, 1, info=info)
    assert len(of) > 500
    # (1) identical field: exact
    info2 = {}
    v2, f2 = meshing.extract(fld['voxel_size'], fld['hier'].levels[0], ev, 1, 1, info=info2)
    ref = pu.ref_from_info(ov, of, info)
    st = pu.compare_meshes('self', v2, f2, info2['vert_vkey'], info2['vert_axis'], ref, delta_f=0.0)
    assert st['exact'] and st['only_hip'] == 0 and np.array_equal(f2, of)
    # (2) perturbed field: a deterministic position-hashed noise, large enough to flip some lattice signs
    fmax = max(float(np.abs(L['f_raw']).max()) for L in info['levels'])
    amp = 2e-3 * fmax

    def noisy(p):
        h = np.sin(p[:, 0] * 12.9898 + p[:, 1] * 78.233 + p[:, 2] * 37.719) * 43758.5453
        return (ev(p) + amp * (h - np.floor(h) - 0.5) * 2).astype(np.float32)
    info3 = {}
    v3, f3 = meshing.extract(fld['voxel_size'], fld['hier'].levels[0], noisy, 1, 1, info=info3)
    delta, _ = pu.lattice_delta(noisy, ref)
    assert 0 < delta <= amp * 1.01
    st = pu.compare_meshes('perturbed', v3, f3, info3['vert_vkey'], info3['vert_axis'], ref, delta_f=delta,
                           max_tainted_frac=1.0, eps=delta * 1.0001, max_over_plain_frac=None)     # tightest valid eps: exactly the noise amplitude
    assert st['only_hip'] + st['only_oracle'] > 0, 'perturbation too small to exercise the tainted-cell logic'
    assert st['outside_tainted'] == 0 and st['tainted_cells'] < 0.5 * st['final_cells']


def test_triangle_cells_recovers_the_emitting_cell():
    from oracle import meshing, pipeline
    fld, xs = _field()
    info = {}
    ov, of = meshing.extract(fld['voxel_size'], fld['hier'].levels[0], lambda p: pipeline.evaluate(fld, p)[0], 0, 1, info=info)
    ids = pu.canonical_triangles(of, info['vert_vkey'], info['vert_axis'])
    lo, hi = pu.triangle_cells(ids)
    tc = info['tri_cell']
    assert ((lo <= tc) & (tc <= hi)).all()
    assert (lo == hi).all(1).mean() > 0.95
