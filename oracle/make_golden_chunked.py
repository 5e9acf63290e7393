"""Generates the committed fixtures of the two chunked BASELINE.json configurations from the CPU oracle.

    python -m oracle.make_golden_chunked [street8|terrain5|all]        (minutes of CPU; run in the dev container)

street8_golden.npz   configs[3] at oracle size: `carla` preset (adaptive_depth 2, UDF mask), SENSOR-ONLY input through
                     the kNN-64 / 85-degree normal recipe (examples/recons_waymo_cpu.py:21-41), chunk_size giving
                     4 x 2 = 8 chunks, approx_kernel_grad=True (examples/recons_waymo.py:30-37), mise_iter=1.
terrain5_golden.npz  configs[4] at oracle size: tree_depth=5, chunked 2 x 2, oriented input, mise_iter=1.
Both hold: the oracle mesh (vertices, faces, canonical vertex ids, |f0-f1| per vertex), the near-threshold lattice
cells needed by tests/parity_util.tainted_cells (cells with a corner |f| < 1e-3 max|f|, with that corner value),
a sample of lattice vertices with their oracle field values (to measure |f_hip - f_oracle|), per-chunk summaries.
The inputs are regenerated from seeds by nksr_amd.utils (numpy RandomState: bit-stable), not stored.
The reference holds no golden vectors for this path: these fixtures pin the HIP path to OUR
oracle on the two configurations that had no oracle comparison in round 1.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, 'tests', 'golden')

CASES = {
    'street8': dict(preset='carla', overrides={}, n=14000, seed=3, chunk_size=3.1, knn=64, deg=85.0, approx=True, tol=3e-7,
                    mise_iter=1),
    'terrain5': dict(preset='ks', overrides={'tree_depth': 5}, n=8000, seed=5, chunk_size=5.0 + 1e-3, knn=None, deg=None, approx=False,
                     tol=3e-7, mise_iter=1),
}


def case_inputs(name):
    """(xyz, normal or None, sensor or None) in model units -- shared with tests/test_gpu_configs.py.
    street8 has no poles: 64 nearest neighbours wrap around a thin cylinder, the two small PCA eigenvalues coincide and the
    "normal" is an arbitrary direction in any implementation (the kept set then differs at the 85-degree cut) -- not a parity case."""
    from nksr_amd import utils
    c = CASES[name]
    if name == 'street8':
        xyz, _, sensor = utils.synth_street(c['n'], seed=c['seed'], extent=(12.0, 6.0), n_boxes=4, n_poles=0, noise=0.002)
        return xyz, None, sensor
    xyz, nrm = utils.synth_terrain_patch(c['n'], seed=c['seed'], extent=(10.0, 10.0))
    return xyz, nrm, None


def pack_mesh_info(prefix, ov, of, info, out, fmax):
    out[prefix + 'v'], out[prefix + 'f'] = ov, of.astype(np.int32)
    out[prefix + 'vert_vkey'], out[prefix + 'vert_axis'] = info['vert_vkey'].astype(np.int64), info['vert_axis'].astype(np.int8)
    out[prefix + 'vert_df'] = info['vert_df'].astype(np.float32)
    out[prefix + 'h'] = np.float64(info['h'])
    out[prefix + 'nlevels'] = np.int64(len(info['levels']))
    rs = np.random.RandomState(0)
    for m, L in enumerate(info['levels']):
        absf = np.abs(L['f'])
        cmin = absf[L['cidx']].min(1) if len(L['cidx']) else np.zeros(0, np.float32)
        near = cmin < 1e-3 * fmax
        out[prefix + 'near_cells_%d' % m] = L['cells'][near].astype(np.int64)
        out[prefix + 'near_minabs_%d' % m] = cmin[near].astype(np.float32)
        out[prefix + 'ncells_%d' % m] = np.int64(len(L['cells']))
        nv = len(L['vk'])
        pick = np.sort(rs.choice(nv, size=min(nv, 20000), replace=False))
        out[prefix + 'probe_vk_%d' % m] = L['vk'][pick].astype(np.int64)
        out[prefix + 'probe_f_%d' % m] = L['f_raw'][pick].astype(np.float32)
        out[prefix + 'lat_h_%d' % m] = np.float64(L['h'])


def run(name):
    import torch
    from nksr_amd import configs
    from nksr_amd.nn.network import NKSRNetwork
    from oracle import chunking, network as onet, normals
    c = CASES[name]
    hp = configs.get_hparams(c['preset'], **c['overrides'])
    P = onet.export_params(NKSRNetwork(hp))
    xyz, nrm, sensor = case_inputs(name)
    pre = None
    if c['knn']:
        def pre(x, n_, s_):
            xs, ns, _, _ = normals.estimate_normals_knn(x, s_, c['knn'], c['deg'])
            return xs, ns, None
    t0 = time.time()
    cf = chunking.reconstruct_by_chunk(xyz, nrm, sensor, c['chunk_size'], preprocess_fn=pre, voxel_size=hp.voxel_size,
                                       depth=hp.tree_depth, adaptive_depth=hp.adaptive_depth, kernel_dim=hp.kernel_dim,
                                       hidden=hp.interpolator.hidden_dim, pos_weight=hp.solver.pos_weight,
                                       normal_weight=hp.solver.normal_weight, tol=c['tol'], approx_kernel_grad=c['approx'],
                                       net_params=P, udf=bool(hp.udf.enabled))
    print(name, 'grid', cf.grid, 'chunks', sorted(cf.fields), 'solve %.1fs' % (time.time() - t0))
    out = {'grid': np.asarray(cf.grid, np.int64), 'chunk_ids': np.asarray(sorted(cf.fields), np.int64), 'ov': np.float64(cf.ov)}
    for k in sorted(cf.fields):
        f = cf.fields[k]
        out['chunk_%d_M' % k] = np.int64(f['A'].shape[0])
        out['chunk_%d_iters' % k] = np.int64(f['iters'])
        out['chunk_%d_nvox' % k] = np.asarray([L.n for L in f['hier'].levels], np.int64)
        out['chunk_%d_keys0' % k] = f['hier'].levels[0].keys
        out['chunk_%d_alpha_absmax' % k] = np.float32(np.abs(f['alpha']).max())
    info = {}
    ov, of = cf.extract_dual_mesh(mise_iter=c['mise_iter'], info=info)
    fmax = max(float(np.abs(L['f_raw']).max()) for L in info['levels'])
    out['fmax'] = np.float64(fmax)
    pack_mesh_info('mesh_', ov, of, info, out, fmax)
    # blended field + gradient at probe points (inputs + jittered inputs)
    rs = np.random.RandomState(1)
    q = xyz[rs.choice(len(xyz), 3000, replace=False)]
    q = np.concatenate([q, q + rs.randn(*q.shape).astype(np.float32) * np.float32(0.05)]).astype(np.float32)
    fq, gq = cf.evaluate(q, grad=True)
    out['probe_xyz'], out['probe_f'], out['probe_grad'] = q, fq, gq
    e = np.sort(np.concatenate([of[:, [0, 1]], of[:, [1, 2]], of[:, [2, 0]]]), 1)
    _, cnt = np.unique(e, axis=0, return_counts=True)
    print(name, 'mesh V=%d F=%d closed=%s fmax=%.4f total %.1fs' % (len(ov), len(of), bool((cnt == 2).all()), fmax, time.time() - t0))
    np.savez_compressed(os.path.join(GOLD, name + '_golden.npz'), **out)
    print('wrote', name, os.path.getsize(os.path.join(GOLD, name + '_golden.npz')) / 1e6, 'MB')


if __name__ == '__main__':
    which = sys.argv[1] if len(sys.argv) > 1 else 'all'
    for nm in CASES:
        if which in ('all', nm):
            run(nm)
