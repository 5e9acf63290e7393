"""Generates the committed fixtures under tests/golden/ from the CPU oracle.

    python -m oracle.make_golden            (needs /root/reference only for the bunny input)

bunny_2k.npz        : 2000-point seeded subsample (RandomState(0)) of the reference asset
                      assets/bunny.ply (xyz + unit normals) -- the only on-disk test input the
                      reference ships.
bunny_2k_golden.npz : oracle outputs for reconstruct(voxel_size=0.05) + extract_dual_mesh on it:
                      voxel keys per level, CSR checksums, alpha, f at the inputs, mesh.
sphere_3k_golden.npz: same for the ShapeNet-3K-noise stand-in (config[1]: sphere r=0.45,
                      sigma=0.005, N=3000, voxel_size 0.05).
The reference has no golden vectors for this path: these pin OUR oracle
against regressions and give the HIP path a fixed target that travels to the GPU box.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, 'tests', 'golden')


def run_case(xyz, nrm, vs):
    from oracle import pipeline
    scale = np.float32(0.1 / vs)
    xs = (xyz * scale).astype(np.float32)
    fld = pipeline.reconstruct(xs, nrm, tol=1e-6)
    f, g = pipeline.evaluate(fld, xs, grad=True)
    out = {'voxel_size': np.float64(vs), 'alpha': fld['alpha'], 'f_at_points': f, 'grad_at_points': g,
           'iters': np.int64(fld['iters']), 'b': fld['b'],
           'A_nnz': np.int64(fld['A'].nnz), 'A_diag': fld['A'].diagonal().astype(np.float32),
           'A_rowsum': np.asarray(fld['A'].sum(1)).ravel().astype(np.float32)}
    for d, L in enumerate(fld['hier'].levels):
        out['keys_%d' % d] = L.keys
    from oracle.make_golden_chunked import pack_mesh_info
    for mise in (0, 1):
        info = {}
        v, t = pipeline.extract_dual_mesh(fld, mise_iter=mise, info=info)
        fmax = max(float(np.abs(L['f_raw']).max()) for L in info['levels'])
        pack_mesh_info('mesh%d_' % mise, v, t, info, out, fmax)      # mesh + what tests/parity_util needs to localise differences
    return out


def main():
    os.makedirs(GOLD, exist_ok=True)
    bunny = os.path.join(GOLD, 'bunny_2k.npz')
    if not os.path.exists(bunny):
        from nksr_amd import utils
        xyz, nrm, _, _ = utils.load_point_cloud('/root/reference/assets/bunny.ply')
        idx = np.sort(np.random.RandomState(0).choice(len(xyz), 2000, replace=False))
        np.savez_compressed(bunny, xyz=xyz[idx], normal=nrm[idx])
    d = np.load(bunny)
    np.savez_compressed(os.path.join(GOLD, 'bunny_2k_golden.npz'), **run_case(d['xyz'], d['normal'], 0.05))
    from nksr_amd import utils
    xyz, nrm = utils.synth_sphere(3000, 0.45, 0.005, seed=0)
    np.savez_compressed(os.path.join(GOLD, 'sphere_3k_golden.npz'), **run_case(xyz, nrm, 0.05))
    print('wrote', sorted(os.listdir(GOLD)))


if __name__ == '__main__':
    main()
