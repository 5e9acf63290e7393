"""Preconditioner policy for SMALL single fields: iterations and end-to-end latency of configs[1] (3 000-point sphere, preset
snet-n3k-wnormal), the 10 000-point bunny scan sequence (examples/recons_waymo_cpu.py:48-63) and the smoke sphere under
Reconstructor.coarse_precond = None (policy) / False (Jacobi) / {'first_level', 'steps', 'ratio'}.
python -m nksr_amd.tools.small_pc_sweep"""
import sys
import time

import numpy as np
import torch

import nksr_amd
from nksr_amd import utils


def lat(fn, reps=6):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, r


def main():
    dev = torch.device('cuda:0')
    import os
    d = np.load(os.path.join(os.path.dirname(__file__), '..', '..', 'tests', 'golden', 'bunny_10k.npz'))

    def synth_sensors(xyz, normal, dist=2.0):      # six scanner positions on the axes; every point is seen from the one its normal faces
        c = xyz.mean(0)
        S = (np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], np.float32) * np.float32(dist) + c).astype(np.float32)
        dd = S[None] - xyz[:, None]
        dd /= np.linalg.norm(dd, axis=2, keepdims=True)
        return S[(dd * normal[:, None]).sum(2).argmax(1)]
    cases = {}
    xyz, nrm = utils.synth_sphere(3000, 0.45, 0.005, 0)
    cases['configs1'] = (nksr_amd.Reconstructor(dev, config='snet-n3k-wnormal'), dict(xyz=torch.from_numpy(xyz).to(dev), normal=torch.from_numpy(nrm).to(dev), detail_level=None))
    bx = torch.from_numpy(d['xyz']).to(dev)
    bs = torch.from_numpy(synth_sensors(d['xyz'], d['normal'])).to(dev)
    cases['bunny'] = (nksr_amd.Reconstructor(dev), dict(xyz=bx, sensor=bs, detail_level=None, approx_kernel_grad=True, solver_tol=1e-4, fused_mode=True,
                                                        preprocess_fn=nksr_amd.get_estimate_normal_preprocess_fn(64, 85.0)))
    xyz, nrm = utils.synth_sphere(2000, 0.45, 0.003, seed=0)
    cases['smoke'] = (nksr_amd.Reconstructor(dev), dict(xyz=torch.from_numpy(xyz).to(dev), normal=torch.from_numpy(nrm).to(dev), voxel_size=0.06, solver_tol=1e-6))
    xyz, nrm = utils.synth_scene(150000, seed=0)
    cases['scene150k'] = (nksr_amd.Reconstructor(dev), dict(xyz=torch.from_numpy(xyz).to(dev), normal=torch.from_numpy(nrm).to(dev), detail_level=1.0))
    settings = [('policy', None), ('jacobi', False)]
    for fl in (1, 2):
        for steps in (4, 6, 8):
            for ratio in (20, 40):
                settings.append(('L%d s%d r%d' % (fl, steps, ratio), {'first_level': fl, 'steps': steps, 'ratio': ratio}))
    for name, (rec, kw) in cases.items():
        xyz = kw.pop('xyz')
        nrm = kw.pop('normal', None)
        for tag, pc in settings:
            rec.coarse_precond = pc
            info = {}

            def seq():
                f = rec.reconstruct(xyz, nrm, **kw)
                info['it'], info['M'], info['rel'] = f.solve_info['iters'], f.solve_info['M'], f.solve_info['rel_residual']
                info['pc'] = (f.solve_info.get('coarse_precond') or {}).get('unknowns')
                return f.extract_dual_mesh(mise_iter=1)
            try:
                ms, m = lat(seq)
                print('%-10s %-12s %7.2f ms  iters %4d  M %7d  block %s  rel %.1e  T %d' % (name, tag, ms, info['it'], info['M'], info['pc'], info['rel'], m.f.shape[0]))
            except Exception as e:
                print('%-10s %-12s FAILED %s' % (name, tag, str(e)[:100]))
            sys.stdout.flush()


if __name__ == '__main__':
    main()
