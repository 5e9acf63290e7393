"""configs[2] step (1M-point cloud, detail_level=1.0, reconstruct + extract_dual_mesh(mise_iter=1)) by itself: wall time per step,
stage times, and -- with --host -- a cProfile of the host side (where the launch train is made).
python -m nksr_amd.tools.prof_cloud [points] [steps] [--host] [--non-fused]"""
import sys
import time

import torch

import nksr_amd
from nksr_amd import utils


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    n = int(args[0]) if args else 1_000_000
    steps = int(args[1]) if len(args) > 1 else 3
    dev = torch.device('cuda:0')
    xyz, nrm = utils.synth_scene(n, seed=0, extent=(40.0, 40.0, 10.0), noise=0.01)
    xyz, nrm = torch.from_numpy(xyz).to(dev), torch.from_numpy(nrm).to(dev)
    rec = nksr_amd.Reconstructor(dev)

    def step():
        f = rec.reconstruct(xyz, nrm, detail_level=1.0, fused_mode='--non-fused' not in sys.argv)
        return f.extract_dual_mesh(mise_iter=1)
    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        m = step()
    torch.cuda.synchronize()
    print('configs[2] %d points: %.2f ms per step, %d triangles' % (n, (time.perf_counter() - t0) / steps * 1e3, m.f.shape[0]))
    if '--host' in sys.argv:
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        pr.disable()
        pstats.Stats(pr).sort_stats('tottime').print_stats(28)


if __name__ == '__main__':
    main()
