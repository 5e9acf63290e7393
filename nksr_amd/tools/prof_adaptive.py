"""The adaptive dual-graph mesher beside the lattice mesher on the configs[2] cloud (for rocprofv3 / kprof.sh).
python -m nksr_amd.tools.prof_adaptive [points] [lattice|adaptive]"""
import sys
import time

import torch

import nksr_amd
from nksr_amd import utils


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    mode = sys.argv[2] if len(sys.argv) > 2 else 'adaptive'
    dev = torch.device('cuda:0')
    xyz, nrm = utils.synth_scene(n, seed=0, extent=(40.0, 40.0, 10.0), noise=0.01)
    rec = nksr_amd.Reconstructor(dev)
    f = rec.reconstruct(torch.from_numpy(xyz).to(dev), torch.from_numpy(nrm).to(dev), detail_level=1.0)
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
