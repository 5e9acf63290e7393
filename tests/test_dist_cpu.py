"""gloo tests (CPU) of the multi-rank protocol -- world size 2, and 4 / 8 ranks on the bench's 8 x 8 chunk layout: chunk partition,
payload exchange, mesh gather + seam merge (nksr_amd/dist.py).  The HIP kernels are not involved -- the GPU side of
the chunked path is covered by tests/test_gpu_chunking.py."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from nksr_amd import dist as D
    try:
        assert D.world() == (rank, world)
        # --- payload exchange: every rank ends up with every chunk, contents intact
        owner = D.partition_chunks(5, world, [50, 10, 40, 30, 20])
        local = {}
        for c in range(5):
            if owner[c] == rank:
                g = torch.Generator().manual_seed(c)
                local[c] = (torch.randint(0, 1 << 40, (100 + 7 * c,), generator=g, dtype=torch.int64),
                            torch.randn(33 * (c + 1), generator=g))
        got = D.exchange_payloads(local, list(range(5)))
        for c in range(5):
            g = torch.Generator().manual_seed(c)
            assert torch.equal(got[c][0], torch.randint(0, 1 << 40, (100 + 7 * c,), generator=g, dtype=torch.int64))
            assert torch.equal(got[c][1], torch.randn(33 * (c + 1), generator=g))
        # --- neighbour-only exchange: a payload goes to the ranks named for it and to nobody else (odd chunks -> the other rank)
        other = 1 - rank
        dest_of = {c: [1 - owner[c]] for c in range(5) if c % 2 == 1}
        got2 = D.exchange_payloads_to(local, dest_of)
        want = sorted(set(local) | {c for c in range(5) if owner[c] == other and c % 2 == 1})
        assert sorted(got2) == want, (sorted(got2), want)
        for c in want:
            g = torch.Generator().manual_seed(c)
            assert torch.equal(got2[c][0], torch.randint(0, 1 << 40, (100 + 7 * c,), generator=g, dtype=torch.int64))
            assert torch.equal(got2[c][1], torch.randn(33 * (c + 1), generator=g))
        # nothing addressed to anybody: the collectives still complete, everybody keeps its own
        assert sorted(D.exchange_payloads_to(local, {})) == sorted(local)
        # the raw all_to_all: ragged lists incl. empty ones, three dtypes
        send = [[torch.arange(3 * r + rank, dtype=torch.int64), torch.full((r,), float(rank)), torch.zeros(2 * rank, dtype=torch.uint8)] for r in range(world)]
        recv = D.all_to_all_tensors(send)
        for r in range(world):
            assert torch.equal(recv[r][0], torch.arange(3 * rank + r, dtype=torch.int64)) and recv[r][1].tolist() == [float(r)] * rank
            assert recv[r][2].numel() == 2 * r and recv[r][2].dtype == torch.uint8
        # --- mesh gather + seam merge: two quads sharing an edge, one per rank
        if rank == 0:
            v = torch.tensor([[0., 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]])
            key = torch.tensor([10, 11, 12, 13])
        else:
            v = torch.tensor([[1., 0, 0], [2, 0, 0], [2, 1, 0], [1, 1, 0]])
            key = torch.tensor([11, 21, 22, 12])
        f = torch.tensor([[0, 1, 2], [0, 2, 3]])
        ax = torch.zeros(4, dtype=torch.int8)
        mv, mf = D.gather_meshes(v, f, key, ax)
        if rank == 0:
            assert mv.shape[0] == 6 and mf.shape[0] == 4
            mfn = mf.numpy()
            e = np.sort(np.concatenate([mfn[:, [0, 1]], mfn[:, [1, 2]], mfn[:, [2, 0]]]), 1)
            _, cnt = np.unique(e, axis=0, return_counts=True)
            assert (cnt == 2).sum() == 3          # two diagonals + the stitched seam edge
        else:
            assert mv.shape[0] == 4 and mf.shape[0] == 2   # non-destination ranks keep their piece
        # --- the same gather with seam candidates: only the flagged vertices (the shared edge's two, + one that is nobody else's) are
        # grouped on rank 0, the others pass through in piece order -- the same mesh in another vertex order
        seam = torch.tensor([0, 1, 1, 0] if rank == 0 else [1, 0, 1, 1], dtype=torch.uint8)
        sv, sf = D.gather_meshes(v, f, key, ax, seam=seam)
        if rank == 0:
            assert sv.shape[0] == 6 and sf.shape[0] == 4
            tri = lambda vv, ff: sorted(tuple(vv[ff[i]].reshape(-1).tolist()) for i in range(ff.shape[0]))
            assert tri(sv, sf) == tri(mv, mf)
        else:
            assert sv.shape[0] == 4 and sf.shape[0] == 2
        # --- the same two quads with five-word vertex names (the adaptive dual graph's pieces: dist.gather_named / merge_named)
        names = torch.stack([torch.ones(4, dtype=torch.int64), key, torch.zeros(4, dtype=torch.int64), torch.zeros(4, dtype=torch.int64), key + 100], 1)
        nv, nf, nn = D.gather_named(v, f, names)
        if rank == 0:
            assert nv.shape[0] == 6 and nf.shape[0] == 4 and nn[:, 1].tolist() == [10, 11, 12, 13, 21, 22]      # name order
            assert torch.equal(nv[nf], torch.cat([torch.tensor([[0., 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]])[f],
                                                  torch.tensor([[1., 0, 0], [2, 0, 0], [2, 1, 0], [1, 1, 0]])[f]]))
        else:
            assert nv.shape[0] == 4 and nf.shape[0] == 2
        # --- an idle rank (more ranks than chunks) owns nothing but still receives everything
        one = {0: (torch.arange(5, dtype=torch.int64), torch.ones(3))} if rank == 1 else {}
        got1 = D.exchange_payloads(one, [0])
        assert sorted(got1) == [0] and torch.equal(got1[0][0], torch.arange(5, dtype=torch.int64)) and got1[0][1].numel() == 3
        # --- empty contribution from a rank still completes the collective
        z = D.all_gather_variable(torch.arange(3 * rank, dtype=torch.int64))
        assert [t.numel() for t in z] == [0, 3]
        q.put((rank, 'ok'))
    except Exception as e:  # surface the failure in the parent
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_two_rank_protocol_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(2)]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, 'ok'), (1, 'ok')], res


def test_partition_is_balanced_and_deterministic():
    from nksr_amd import dist as D
    w = [100, 90, 80, 10, 10, 10, 5, 5]
    o = D.partition_chunks(len(w), 4, w)
    assert o == D.partition_chunks(len(w), 4, w)
    load = [sum(w[c] for c in range(len(w)) if o[c] == r) for r in range(4)]
    assert max(load) <= 100 and min(load) >= 30
    assert D.partition_chunks(3, 8) == [0, 1, 2]          # more ranks than chunks: idle ranks
    from nksr_amd.chunking import needed_chunks
    cores = {c: ([10.0 * c, 0, 0], [10.0 * c + 10, 10, 10]) for c in range(6)}
    assert needed_chunks(cores, 0.9, [6, 1, 1], [2], list(range(6))) == [1, 2, 3]
    assert needed_chunks(cores, 0.9, [6, 1, 1], [0, 5], list(range(6))) == [0, 1, 4, 5]
    assert D.merge_meshes([(torch.zeros((0, 3)), torch.zeros((0, 3), dtype=torch.int64), torch.zeros(0, dtype=torch.int64),
                            torch.zeros(0, dtype=torch.int8))])[0].shape[0] == 0


# ---- the layout of the multi-GPU bench (8 x 8 tiles, Morton-cut partition) at 4 and 8 ranks ------------------------------------------
def _tile_layout(world):
    from nksr_amd import dist as D
    grid = [8, 8, 1]
    rs = np.random.RandomState(7)
    counts = [int(v) for v in rs.randint(100_000, 220_000, 64)]
    counts[5] = counts[40] = 0                                           # two tiles without points: never solved, never sent
    cores = {c: ([125.0 * (c // 8), 125.0 * (c % 8), 0.0], [125.0 * (c // 8) + 125.0, 125.0 * (c % 8) + 125.0, 30.0]) for c in range(64)}
    owner = D.partition_chunks(64, world, counts, grid)
    return grid, counts, cores, owner


def _payload_of(c):
    g = torch.Generator().manual_seed(1000 + c)
    return (torch.randint(0, 1 << 40, (50 + 3 * c,), generator=g, dtype=torch.int64), torch.randn(20 + c, generator=g))


def _tile_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from nksr_amd import dist as D
    from nksr_amd.chunking import halo_destinations, needed_chunks
    try:
        grid, counts, cores, owner = _tile_layout(world)
        margin = 12.5 + 2.5 * 0.1
        mine = [c for c in range(64) if owner[c] == rank and counts[c] > 0]
        local = {c: _payload_of(c) for c in mine}
        dest_of = halo_destinations(cores, margin, grid, owner, counts, world)
        got = D.exchange_payloads_to(local, dest_of)
        # exactly the chunks whose blend support reaches this rank's cores: its own + their spatial neighbours on other ranks
        nonempty = [c for c in range(64) if counts[c] > 0]
        want = needed_chunks(cores, margin, grid, mine, nonempty)
        assert sorted(got) == want, (rank, sorted(got), want)
        for c in want:
            a, b = _payload_of(c)
            assert torch.equal(got[c][0], a) and torch.equal(got[c][1], b), (rank, c)
        # a rank never receives more than its 8-neighbourhood ring: far fewer than all 64 halos
        assert len(want) - len(mine) <= 64 - len(mine) and (world == 2 or len(want) < 48)
        # mesh gather: a strip of quads, one per rank, each sharing an edge with the next rank's
        v = torch.tensor([[rank, 0., 0], [rank + 1, 0, 0], [rank + 1, 1, 0], [rank, 1, 0]], dtype=torch.float32)
        key = torch.tensor([2 * rank, 2 * rank + 2, 2 * rank + 3, 2 * rank + 1])
        f = torch.tensor([[0, 1, 2], [0, 2, 3]])
        mv, mf = D.gather_meshes(v, f, key, torch.zeros(4, dtype=torch.int8))
        if rank == 0:
            assert mv.shape[0] == 2 * world + 2 and mf.shape[0] == 2 * world, (mv.shape, mf.shape)
            mfn = mf.numpy()
            e = np.sort(np.concatenate([mfn[:, [0, 1]], mfn[:, [1, 2]], mfn[:, [2, 0]]]), 1)
            _, cnt = np.unique(e, axis=0, return_counts=True)
            assert (cnt == 2).sum() == world + (world - 1)          # one diagonal per quad + every stitched seam
        q.put((rank, 'ok'))
    except Exception as e:
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def _run_world(target, world):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(r, 'ok') for r in range(world)], res


def test_bench_tile_layout_at_four_and_eight_ranks_gloo():
    """The 8 x 8 tiles of the multi-GPU bench cut for 4 and 8 ranks: every rank gets exactly the halos its cores read (its spatial
    neighbours on other ranks), intact, and nothing else; the strip of per-rank meshes is stitched at every seam."""
    for world in (4, 8):
        _run_world(_tile_worker, world)


def test_halo_destinations_are_symmetric_neighbours_only():
    from nksr_amd.chunking import halo_destinations
    for world in (2, 4, 8):
        grid, counts, cores, owner = _tile_layout(world)
        assert sorted(set(owner)) == list(range(world))
        load = [sum(counts[c] for c in range(64) if owner[c] == r) for r in range(world)]
        assert max(load) <= 1.25 * min(load), load                     # Morton cut: balanced to within a tile
        dest = halo_destinations(cores, 12.75, grid, owner, counts, world)
        assert 5 not in dest and 40 not in dest                         # empty tiles are never sent
        for c, ranks in dest.items():
            assert owner[c] not in ranks and len(set(ranks)) == len(ranks)
            for r in ranks:                                             # r owns a tile adjacent (incl. diagonally) to c
                assert any(counts[o] > 0 and owner[o] == r and abs(o // 8 - c // 8) <= 1 and abs(o % 8 - c % 8) <= 1 for o in range(64)), (c, r)
        sent = sum(len(v) for v in dest.values())
        assert sent < 62 * (world - 1) * 0.75                           # far fewer than every halo to every other rank
