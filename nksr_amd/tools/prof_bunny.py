"""Latency of the examples/recons_waymo_cpu.py call sequence on the 10 000-point bunny scan and of configs[1] (3 000 points):
warm wall time per call, stage split, host-side hot spots.   python -m nksr_amd.tools.prof_bunny [reps]"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

import nksr_amd
from nksr_amd import configs, utils

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    dev = torch.device('cuda:0')
    d = np.load(os.path.join(ROOT, 'tests', 'golden', 'bunny_10k.npz'))        # assets/bunny.ply of the reference, as committed
    bx = torch.from_numpy(d['xyz']).to(dev)
    # six scanners on the axes around the centroid, every point seen from the one its normal faces (as bench.py's cpu_baseline leg)
    cen = d['xyz'].mean(0)
    S = (np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], np.float32) * np.float32(2.0) + cen).astype(np.float32)
    dd = S[None] - d['xyz'][:, None]
    dd /= np.linalg.norm(dd, axis=2, keepdims=True)
    bs = torch.from_numpy(S[(dd * d['normal'][:, None]).sum(2).argmax(1)]).to(dev)
    rec = nksr_amd.Reconstructor(dev)
    fn = nksr_amd.get_estimate_normal_preprocess_fn(64, 85.0)

    def bunny():
        f = rec.reconstruct(bx, sensor=bs, detail_level=None, approx_kernel_grad=True, solver_tol=1e-4, fused_mode=True, preprocess_fn=fn)
        return f, f.extract_dual_mesh(mise_iter=1)
    xyz, nrm = utils.synth_sphere(3000, 0.45, 0.005, seed=0)
    sx, sn = torch.from_numpy(xyz).to(dev), torch.from_numpy(nrm).to(dev)
    rec1 = nksr_amd.Reconstructor(dev, hparams=configs.get_hparams('snet-n3k-wnormal'))

    def shapenet():
        f = rec1.reconstruct(sx, sn, detail_level=None)
        return f, f.extract_dual_mesh(mise_iter=1)
    for name, fn_, r in (('bunny 10k (recons_waymo_cpu sequence)', bunny, rec), ('configs[1] 3k (snet-n3k-wnormal)', shapenet, rec1)):
        for _ in range(3):
            fn_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            f, m = fn_()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print('%s: %.2f ms per call, iters %s, M %s, T %d' % (name, dt * 1e3, f.solve_info.get('iters'), f.solve_info.get('M'), m.f.shape[0]))
        r.sync_timing = True
        f, m = fn_()
        print('   stages (synchronised):', {k: round(v * 1e3, 2) for k, v in r.timing.items()})
        r.sync_timing = False
        if os.environ.get('NKSR_CPROFILE'):
            pr = cProfile.Profile()
            pr.enable()
            for _ in range(reps):
                fn_()
            torch.cuda.synchronize()
            pr.disable()
            pstats.Stats(pr).sort_stats('tottime').print_stats(22)


if __name__ == '__main__':
    main()
