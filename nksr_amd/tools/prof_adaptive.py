"""The adaptive dual-graph mesher beside the lattice mesher on the configs[2] cloud (for rocprofv3 / kprof.sh).
python -m nksr_amd.tools.prof_adaptive [points] [lattice|adaptive] [chunk_size: a chunked field in one process]"""
import sys
import time

import torch

import nksr_amd
from nksr_amd import utils


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    mode = sys.argv[2] if len(sys.argv) > 2 else 'adaptive'
    dev = torch.device('cuda:0')
    if len(sys.argv) > 3 and sys.argv[3] == 'scene':          # the bench's configs[4] scene (64 chunks in one process), solved once
        import os
        sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.getcwd()))
        import bench
        from nksr_amd import configs
        rec = nksr_amd.Reconstructor(dev, hparams=configs.get_hparams('ks', tree_depth=5))
        xyz, nrm, scale, owner, bounds, _, _ = bench.terrain_setup(rec, dev, n, 0, 1)
        f = rec.reconstruct(xyz, nrm, detail_level=None, chunk_size=bench.TILE * scale, sharded_input=True, chunk_owner=owner, chunk_bounds=bounds)
        f.dual_graph = mode
        m = f.extract_dual_mesh(mise_iter=1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            m = f.extract_dual_mesh(mise_iter=1)
        torch.cuda.synchronize()
        print('scene %s: %.2f ms, V=%d T=%d' % (mode, (time.perf_counter() - t0) / 3 * 1e3, m.v.shape[0], m.f.shape[0]))
        return
    xyz, nrm = utils.synth_scene(n, seed=0, extent=(40.0, 40.0, 10.0), noise=0.01)
    rec = nksr_amd.Reconstructor(dev)
    chunk = float(sys.argv[3]) if len(sys.argv) > 3 else -1
    f = rec.reconstruct(torch.from_numpy(xyz).to(dev), torch.from_numpy(nrm).to(dev), detail_level=(None if chunk > 0 else 1.0), chunk_size=chunk)
    f.dual_graph = mode
    for _ in range(2):
        m = f.extract_dual_mesh(mise_iter=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        m = f.extract_dual_mesh(mise_iter=1)
    torch.cuda.synchronize()
    print('%s: %.2f ms, V=%d T=%d' % (mode, (time.perf_counter() - t0) / 3 * 1e3, m.v.shape[0], m.f.shape[0]))


if __name__ == '__main__':
    main()
