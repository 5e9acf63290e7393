"""Per-rank cost of the sharded (N > 1) bench path, simulated in ONE process on one GPU: the compute a
rank does (select + solve its chunk, pack, unpack the neighbours' payloads, mesh its cells, rank 0's
seam merge) is timed; the collectives themselves are not (payload sizes are printed instead).
python -m nksr_amd.tools.prof_dist [world] [points per rank]"""
import sys
import time

import numpy as np
import torch

import nksr_amd
from nksr_amd import chunking, dist, meshing, utils
from nksr_amd.density import scale_for_detail_level


def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
    dev = torch.device('cuda:0')
    extent = (40.0, 40.0, 10.0)
    rec = nksr_amd.Reconstructor(dev)
    tiles = [utils.synth_scene(n, seed=r, extent=extent, noise=0.01, origin=(r * extent[0], 0.0, 0.0)) for r in range(world)]
    scale = scale_for_detail_level(torch.from_numpy(tiles[0][0]).to(dev), 1.0, rec.hparams.voxel_size)
    xyz_np = np.concatenate([t[0] for t in tiles]) * np.float32(scale)
    xyz_np[:, 0] -= xyz_np[:, 0].min()
    xyz = torch.from_numpy(xyz_np.astype(np.float32)).to(dev)
    nrm = torch.from_numpy(np.concatenate([t[1] for t in tiles])).to(dev)
    chunk_size = float(xyz_np[:, 0].max()) / world + 1e-3
    args = (rec, xyz, nrm, None, chunk_size, 0.05, False, 2000, 1e-5, True, None)

    def sync():
        torch.cuda.synchronize()
        return time.perf_counter()

    me = min(1, world - 1)                       # a rank with neighbours on both sides when world > 2
    for rep in range(2):                         # second pass is the warm one
        t0 = sync()
        mine = chunking.reconstruct_by_chunk(*args, sim=(me, world))
        t1 = sync()
        payload = {}
        for c, f in mine.fields.items():
            c3 = (c // (mine.grid[1] * mine.grid[2]), (c // mine.grid[2]) % mine.grid[1], c % mine.grid[2])
            payload[c] = chunking.pack_field(f, chunking.exchange_band(mine.cores[c], c3, mine.grid, mine.ov, rec.hparams.voxel_size))
        t2 = sync()
    others = {}
    for r in range(world):
        if r != me:
            for c, f in chunking.reconstruct_by_chunk(*args, sim=(r, world)).fields.items():
                c3 = (c // (mine.grid[1] * mine.grid[2]), (c // mine.grid[2]) % mine.grid[1], c % mine.grid[2])
                others[c] = chunking.pack_field(f, chunking.exchange_band(mine.cores[c], c3, mine.grid, mine.ov, rec.hparams.voxel_size))
    nonempty = sorted(list(payload) + list(others))
    owned = list(payload)
    need = chunking.needed_chunks(mine.cores, mine.ov + rec.hparams.voxel_size, mine.grid, owned, nonempty)
    for rep in range(2):
        t3 = sync()
        fields = {c: (mine.fields[c] if c in mine.fields else
                      chunking.unpack_field(others[c][0], others[c][1], rec.hparams.voxel_size, rec.network.interpolators, dev)) for c in need}
        mf = mine.for_rank(me, world, fields)
        t4 = sync()
        piece = meshing._extract(mf, 1, 1, -1)
        t5 = sync()
    print('world=%d rank=%d: select+solve %.1f ms | pack %.1f ms | unpack %d neighbour chunk(s) + union grid %.1f ms | mesh (owned cells + halo) %.1f ms'
          % (world, me, (t1 - t0) * 1e3, (t2 - t1) * 1e3, len(need) - len(owned), (t4 - t3) * 1e3, (t5 - t4) * 1e3))
    pi, pf = payload[owned[0]]
    print('payload per rank: %.1f MB ints + %.1f MB floats; mesh piece: V=%d F=%d (%.1f MB)' % (
        pi.numel() * 8 / 1e6, pf.numel() * 4 / 1e6, piece.v.shape[0], piece.f.shape[0],
        (piece.v.numel() * 4 + piece.f.numel() * 8 + piece.edge_vkey.numel() * 16) / 1e6))
    # rank 0's merge of `world` pieces of this size
    pieces = [(piece.v, piece.f, piece.edge_vkey + r * 7919, piece.edge_axis) for r in range(world)]
    for rep in range(2):
        t6 = sync()
        v, f = dist.merge_meshes(pieces)
        t7 = sync()
    print('rank-0 merge of %d pieces: %.1f ms  (V=%d F=%d)' % (world, (t7 - t6) * 1e3, v.shape[0], f.shape[0]))


if __name__ == '__main__':
    main()
