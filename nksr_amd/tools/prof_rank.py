"""What ONE rank of an N-rank run of the configs[4] bench scene computes, timed on one GPU without a process group: the tiles of the
rank's chunks (+ neighbours), the batched solve of its chunks (reconstruct_by_chunk(sim=(rank, world))).  The halo exchange and the
mesh gather are not part of it.   python -m nksr_amd.tools.prof_rank [world] [rank] [scene points]"""
import sys
import time

import torch

import bench
import nksr_amd
from nksr_amd import chunking, configs


def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    rank = int(sys.argv[2]) if len(sys.argv) > 2 else world // 2
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 10_000_000
    dev = torch.device('cuda:0')
    rec = nksr_amd.Reconstructor(dev, hparams=configs.get_hparams('ks', tree_depth=5))
    rec.sync_timing = True
    xyz, nrm, scale, owner, bounds, n_scene, ntiles = bench.terrain_setup(rec, dev, n, rank, world)
    cs = bench.TILE * scale
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        f = chunking.reconstruct_by_chunk(rec, xyz, nrm, None, cs, 0.05, False, 2000, 1e-5, True, None, sim=(rank, world), sharded_input=True,
                                          chunk_owner=owner, chunk_bounds=bounds)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        print('world %d rank %d: %d chunks, %d points resident, solve %.1f ms  %s' % (
            world, rank, sum(1 for o in owner if o == rank), xyz.shape[0], (t1 - t0) * 1e3,
            {k: round(v * 1e3, 1) for k, v in rec.timing.items()} if getattr(rec, 'timing', None) else ''))
        del f


if __name__ == '__main__':
    main()
