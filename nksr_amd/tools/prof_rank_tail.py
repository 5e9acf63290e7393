"""Everything ONE rank of an N-rank run of the configs[4] bench scene does, timed on one GPU without a process group: the batched
solve of its chunks, the halo exchange step with the collectives replaced by a dictionary (pack, the remote field's tables, the union
grid), the meshing of its cells -- and rank 0's seam merge of N pieces of that size.  The neighbours' halos come from simulated
runs of their ranks (untimed).   python -m nksr_amd.tools.prof_rank_tail [world] [rank] [scene points]"""
import sys
import time

import torch

import bench
import nksr_amd
from nksr_amd import chunking, configs, dist


def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    rank = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 10_000_000
    dev = torch.device('cuda:0')
    rec = nksr_amd.Reconstructor(dev, hparams=configs.get_hparams('ks', tree_depth=5))
    import os
    graph = os.environ.get('NKSR_TAIL_GRAPH', 'lattice')      # 'adaptive': deeper halos, cell-pair vertex names, dist.merge_named
    rec.dual_graph = graph
    sent = {}            # chunk -> (payload, destination ranks): what every simulated rank packs

    inputs = {}

    def run(r, exchange):
        if r not in inputs:
            inputs.clear()                                   # (one rank's tiles resident at a time)
            inputs[r] = bench.terrain_setup(rec, dev, n, r, world)
        xyz, nrm, scale, owner, bounds, n_scene, ntiles = inputs[r]
        return chunking.reconstruct_by_chunk(rec, xyz, nrm, None, bench.TILE * scale, 0.05, False, 2000, 1e-5, True, None, sim=(r, world),
                                             sharded_input=True, chunk_owner=owner, chunk_bounds=bounds, sim_exchange=exchange), owner

    def record(local, dest_of):
        for c, p in local.items():
            sent[c] = (p, dest_of.get(c, []))
        return dict(local)

    # pass 1: this rank alone tells which ranks send to it (dest_of is the same table on every rank)
    dests = {}

    def probe(local, dest_of):
        dests.update(dest_of)
        return dict(local)
    _, owner = run(rank, probe)
    senders = sorted({owner[c] for c, rs in dests.items() if rank in rs})
    print('world %d rank %d: receives halos from ranks %s' % (world, rank, senders))
    for r in senders:
        run(r, record)
    torch.cuda.empty_cache()

    def mine(local, dest_of):
        out = dict(local)
        for c, (p, rs) in sent.items():
            if rank in rs:
                out[c] = p
        return out
    rec.sync_timing = True
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        f, _ = run(rank, mine)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        res = f.extract_dual_mesh(mise_iter=1)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        tm = {k: round(v * 1e3, 1) for k, v in rec.timing.items()}
        print('rep %d: setup + solve + exchange step + union grid %.1f ms %s | mesh of own cells %.1f ms (V=%d T=%d)' % (
            rep, (t1 - t0) * 1e3, tm, (t2 - t1) * 1e3, res.v.shape[0], res.f.shape[0]))
    if os.environ.get('NKSR_CPROFILE'):
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        for rep in range(3):
            f, _ = run(rank, mine)
            res = f.extract_dual_mesh(mise_iter=1)
        torch.cuda.synchronize()
        pr.disable()
        pstats.Stats(pr).sort_stats('cumulative').print_stats(45)
    if os.environ.get('NKSR_TAIL_STEPS_ONLY'):      # (for a kernel trace whose last milliseconds are one rank's step)
        return
    recv = [c for c, (p, rs) in sent.items() if rank in rs]
    print('halos received: %d chunks, %.2f MB' % (len(recv), sum(sent[c][0][0].numel() * 8 + sent[c][0][1].numel() * 4 for c in recv) / 1e6))
    if graph == 'adaptive':
        def shifted(r):
            nm = res.vertex_names5.clone()
            nm[:, 1] += r * 7919
            return nm
        pieces = [(res.v, res.f, shifted(r)) for r in range(world)]
    else:
        fl = getattr(res, 'seam_flag', None)      # (the pieces say which vertices another rank may hold too: rank 0 groups only those)
        pieces = [(res.v, res.f, res.edge_vkey + r * 7919, res.edge_axis) + ((fl,) if fl is not None else ()) for r in range(world)]
        print('seam candidates: %s of %d vertices per piece' % (int(fl.sum()) if fl is not None else 'all', res.v.shape[0]))
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        v, ff = (dist.merge_named(pieces)[:2] if graph == 'adaptive' else dist.merge_meshes(pieces))
        torch.cuda.synchronize()
        print('rank-0 merge of %d pieces (%s): %.1f ms (V=%d T=%d, %.0f MB gathered)' % (
            world, graph, (time.perf_counter() - t0) * 1e3, v.shape[0], ff.shape[0],
            (world - 1) * (res.v.numel() * 4 + res.f.numel() * 8 + res.v.shape[0] * (40 if graph == 'adaptive' else 9)) / 1e6))


if __name__ == '__main__':
    main()
