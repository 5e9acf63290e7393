"""Multi-GPU glue for the chunked path: one process per GPU,
``torch.distributed`` (backend "nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The solve of a chunk is independent of every other chunk (the reference runs chunks sequentially
on one device, examples/recons_by_chunk.py:26-29), so chunks are sharded over ranks with NO
collective on the solve path.  Exactly one exchange step precedes meshing -- every rank needs the
solved fields that overlap the cells it meshes -- and one gather step follows it:
  * exchange_payloads_to: ONE all_to_all_single of sizes (the only host sync) + ONE all_to_all_single of a byte buffer with
    per-pair split sizes: a rank receives the halos of the chunks whose blend weight reaches its cells -- its spatial
    neighbours' -- and nothing else (round 3: an all_gather gave every halo to every rank); on the fully connected xGMI
    mesh these are direct peer copies, no ring
  * gather_meshes: one size collective, then point-to-point transfers to rank 0 only (the other ranks
    receive nothing); rank 0 merges seam vertices by their canonical (lattice key, axis) identity.
Everything here works on CPU tensors too, so the protocol is covered by world_size-2 gloo tests.
"""
import torch


def is_dist():
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


def world():
    import torch.distributed as dist
    return (dist.get_rank(), dist.get_world_size()) if is_dist() else (0, 1)


def active():
    """True when the collectives below really run: a process group of more than one rank -- or of ONE rank with
    NKSR_DIST_FORCE=1, which pushes the whole exchange / gather path through the backend (RCCL on a single GPU: the
    first time RCCL touches these buffers must not be an 8-GPU run)."""
    import os
    if not is_dist():
        return False
    return world()[1] > 1 or os.environ.get('NKSR_DIST_FORCE', '') == '1'


def _morton3(x, y, z):
    k = 0
    for b in range(21):
        k |= ((x >> b) & 1) << (3 * b) | ((y >> b) & 1) << (3 * b + 1) | ((z >> b) & 1) << (3 * b + 2)
    return k


def partition_chunks(n_chunks, world_size, weights=None, grid=None):
    """owner[c] for every chunk.  Chunks are ordered along a Morton curve over their grid position (``grid`` = chunks per
    axis, chunk id = (cx * grid[1] + cy) * grid[2] + cz; plain index order without it) and the curve is cut into
    ``world_size`` contiguous pieces of (nearly) equal total ``weights`` (point counts): a rank's chunks are spatially
    compact, so it shares halos with few other ranks and -- with sharded input -- loads few extra tiles.  Pure integer
    arithmetic => identical on every rank."""
    if weights is None:
        return [c % world_size for c in range(n_chunks)]
    if grid is not None:
        g1, g2 = int(grid[1]), int(grid[2])
        order = sorted(range(n_chunks), key=lambda c: (_morton3(c // (g1 * g2), (c // g2) % g1, c % g2), c))
    else:
        order = list(range(n_chunks))
    w = [max(int(weights[c]), 0) for c in range(n_chunks)]
    total = sum(w)
    owner = [0] * n_chunks
    if total == 0:
        return [c % world_size for c in range(n_chunks)]
    acc = 0
    for c in order:
        # rank of the piece that holds the midpoint of this chunk's weight interval
        owner[c] = min(world_size - 1, ((2 * acc + w[c]) * world_size) // (2 * total))
        acc += w[c]
    return owner


def _comm_device(t):
    import torch.distributed as dist
    return t.device if dist.get_backend() != 'gloo' else torch.device('cpu')


def _default_device():
    import torch.distributed as dist
    return torch.device('cpu') if dist.get_backend() == 'gloo' else torch.device('cuda', torch.cuda.current_device())


def _pack_bytes(tensors):
    """list of tensors (any dtype) -> (uint8 buffer, [(dtype, numel)])."""
    parts = [t.contiguous().reshape(-1).view(torch.uint8) for t in tensors]
    pad = [torch.zeros((-p.numel()) % 8, dtype=torch.uint8, device=p.device) for p in parts]     # keep every part 8-byte aligned
    buf = torch.cat([x for pp in zip(parts, pad) for x in pp]) if parts else torch.zeros(0, dtype=torch.uint8)
    return buf, [(t.dtype, t.numel()) for t in tensors]


def _unpack_bytes(buf, dtypes, numels):
    out, o = [], 0
    for dt, n in zip(dtypes, numels):
        nb = n * torch.empty(0, dtype=dt).element_size()
        out.append(buf[o:o + nb].view(dt))
        o += nb + ((-nb) % 8)
    return out


def all_gather_tensors(tensors):
    """all_gather of a LIST of 1-D tensors whose lengths differ per rank (dtypes are the same on every rank): ONE size
    collective ([k] int64 per rank, the only host sync) + ONE padded byte collective.  Returns per rank the list of
    tensors, on the device of the inputs."""
    import torch.distributed as dist
    rank, ws = world()
    if not active():
        return [list(tensors)]
    src_dev = tensors[0].device
    dev = _comm_device(tensors[0])
    buf, meta = _pack_bytes([t.to(dev) for t in tensors])
    k = len(tensors)
    n = torch.tensor([m[1] for m in meta], dtype=torch.int64, device=dev)
    sizes = torch.empty(ws * k, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(sizes, n) if dist.get_backend() != 'gloo' else dist.all_gather(list(sizes.view(ws, k).unbind(0)), n)
    sizes = sizes.view(ws, k).tolist()                       # one host sync for the whole exchange
    es = [torch.empty(0, dtype=m[0]).element_size() for m in meta]
    nbytes = [sum(s * e + ((-(s * e)) % 8) for s, e in zip(row, es)) for row in sizes]
    mx = max(max(nbytes), 8)
    send = torch.zeros(mx, dtype=torch.uint8, device=dev)
    send[:buf.numel()] = buf
    out = torch.empty(ws * mx, dtype=torch.uint8, device=dev)
    if dist.get_backend() != 'gloo':
        dist.all_gather_into_tensor(out, send)
    else:
        dist.all_gather(list(out.view(ws, mx).unbind(0)), send)
    res = []
    for r in range(ws):
        res.append([t.to(src_dev) for t in _unpack_bytes(out[r * mx:r * mx + nbytes[r]], [m[0] for m in meta], sizes[r])])
    return res


def all_to_all_tensors(send):
    """``send[r]`` = the list of k 1-D tensors this rank has for rank r (same k and dtypes for every pair; any lengths, empty
    allowed).  Returns ``recv[r]`` = the k tensors rank r had for this rank.  Two collectives: ONE all_to_all_single of the k
    lengths per pair (the only host sync) + ONE all_to_all_single of a packed byte buffer with per-pair split sizes -- on the fully
    connected xGMI mesh RCCL runs that as direct peer copies, and a rank only receives what was addressed to it (the all_gather
    of round 3 delivered every halo to every rank)."""
    import torch.distributed as dist
    rank, ws = world()
    if not active():
        return [list(send[0])]
    k = len(send[0])
    src_dev = send[0][0].device
    dev = _comm_device(send[0][0])
    dtypes = [t.dtype for t in send[0]]
    packed = [_pack_bytes([t.to(dev) for t in send[r]])[0] for r in range(ws)]
    n_out = torch.tensor([[t.numel() for t in send[r]] for r in range(ws)], dtype=torch.int64, device=dev).reshape(-1)
    n_in = torch.empty(ws * k, dtype=torch.int64, device=dev)
    dist.all_to_all_single(n_in, n_out)
    sizes = n_in.view(ws, k).tolist()                         # one host sync for the whole exchange
    es = [torch.empty(0, dtype=dt).element_size() for dt in dtypes]
    in_bytes = [sum(n * e + ((-(n * e)) % 8) for n, e in zip(row, es)) for row in sizes]
    out_bytes = [int(p.numel()) for p in packed]
    inp = torch.cat(packed) if sum(out_bytes) else torch.zeros(0, dtype=torch.uint8, device=dev)
    out = torch.empty(sum(in_bytes), dtype=torch.uint8, device=dev)
    dist.all_to_all_single(out, inp, output_split_sizes=in_bytes, input_split_sizes=out_bytes)
    recv, o = [], 0
    for r in range(ws):
        recv.append([t.to(src_dev) for t in _unpack_bytes(out[o:o + in_bytes[r]], dtypes, sizes[r])])
        o += in_bytes[r]
    return recv


def all_gather_variable(t):
    """all_gather of 1-D tensors whose lengths differ per rank.  Returns the list of per-rank tensors (on t's device)."""
    return [r[0] for r in all_gather_tensors([t])]


def gather_tensors(tensors, dst=0):
    """Variable-length gather of a list of 1-D tensors to rank ``dst`` ONLY: one size collective, then point-to-point
    transfers (batched isend / irecv) -- the other ranks receive nothing.  Returns the per-rank lists on ``dst``, None elsewhere."""
    import torch.distributed as dist
    rank, ws = world()
    if not active():
        return [list(tensors)]
    src_dev = tensors[0].device
    dev = _comm_device(tensors[0])
    buf, meta = _pack_bytes([t.to(dev) for t in tensors])
    k = len(tensors)
    n = torch.tensor([m[1] for m in meta], dtype=torch.int64, device=dev)
    sizes = torch.empty(ws * k, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(sizes, n) if dist.get_backend() != 'gloo' else dist.all_gather(list(sizes.view(ws, k).unbind(0)), n)
    sizes = sizes.view(ws, k).tolist()
    es = [torch.empty(0, dtype=m[0]).element_size() for m in meta]
    nbytes = [sum(s * e + ((-(s * e)) % 8) for s, e in zip(row, es)) for row in sizes]
    if rank != dst:
        if nbytes[rank]:
            for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, buf, dst)]):
                w.wait()
        return None
    bufs = [buf if r == dst else torch.empty(nbytes[r], dtype=torch.uint8, device=dev) for r in range(ws)]
    ops = [dist.P2POp(dist.irecv, bufs[r], r) for r in range(ws) if r != dst and nbytes[r]]
    import os
    if ws == 1 and os.environ.get('NKSR_DIST_FORCE', '') == '1' and nbytes[dst]:
        # single-rank group forced through the backend (tests/test_gpu_rccl1.py): the destination's own piece takes the point-to-point
        # transport too -- a send to self inside one group call -- so that batch_isend_irecv has run under RCCL before an 8-GPU run
        bufs[dst] = torch.empty_like(buf)
        ops = [dist.P2POp(dist.isend, buf, dst), dist.P2POp(dist.irecv, bufs[dst], dst)]
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return [[t.to(src_dev) for t in _unpack_bytes(bufs[r], [m[0] for m in meta], sizes[r])] for r in range(ws)]


def exchange_payloads_to(local, dest_of):
    """Neighbour-only halo exchange: ``local`` = {chunk_id: (int64 tensor, float32 tensor)} of the chunks this rank solved,
    ``dest_of[chunk]`` = the ranks whose cells that chunk's blend weight reaches (chunking.needed_chunks; the owner excluded).
    Every rank receives exactly the payloads addressed to it: returns {chunk_id: (ints, floats)} = own chunks + received ones.
    One size collective + one byte collective (all_to_all_tensors); every rank takes part, with or without chunks."""
    rank, ws = world()
    if not active():
        return dict(local)
    ids = sorted(local)
    dev = local[ids[0]][0].device if ids else _default_device()
    send = []
    for r in range(ws):
        mine = [c for c in ids if r != rank and r in dest_of.get(c, ())]
        if ws == 1:
            mine = ids                                              # (forced single-rank group: everything goes through the backend once)
        head = torch.tensor([v for c in mine for v in (c, local[c][0].numel(), local[c][1].numel())], dtype=torch.int64, device=dev)
        ib = torch.cat([local[c][0].reshape(-1) for c in mine]) if mine else torch.zeros(0, dtype=torch.int64, device=dev)
        fb = torch.cat([local[c][1].reshape(-1) for c in mine]) if mine else torch.zeros(0, dtype=torch.float32, device=dev)
        send.append([head, ib, fb])
    out = dict(local)
    for h, ib, fb in all_to_all_tensors(send):
        io = fo = 0
        hl = h.tolist()
        for k in range(0, len(hl), 3):
            c, ni, nf = hl[k], hl[k + 1], hl[k + 2]
            out[c] = (ib[io:io + ni], fb[fo:fo + nf])
            io += ni
            fo += nf
    return out


def exchange_payloads(local, expected_ids=None):
    """``local``: {chunk_id: (int64 tensor, float32 tensor)} for the chunks this rank owns.
    Returns the same dict for ALL chunks on every rank (one size collective + one byte collective): the everyone-gets-everything
    variant (save / load of whole scenes, tests); the meshing path uses exchange_payloads_to."""
    rank, ws = world()
    if not active():
        return dict(local)
    ids = sorted(local)
    dev = local[ids[0]][0].device if ids else _default_device()     # an idle rank still takes part in the collectives
    head = torch.tensor([v for c in ids for v in (c, local[c][0].numel(), local[c][1].numel())], dtype=torch.int64, device=dev)
    ibuf = torch.cat([local[c][0].reshape(-1) for c in ids]) if ids else torch.zeros(0, dtype=torch.int64, device=dev)
    fbuf = torch.cat([local[c][1].reshape(-1) for c in ids]) if ids else torch.zeros(0, dtype=torch.float32, device=dev)
    out = {}
    for h, ib, fb in all_gather_tensors([head, ibuf, fbuf]):
        io = fo = 0
        hl = h.tolist()
        for k in range(0, len(hl), 3):
            c, ni, nf = hl[k], hl[k + 1], hl[k + 2]
            out[c] = (ib[io:io + ni], fb[fo:fo + nf])
            io += ni
            fo += nf
    if expected_ids is not None:
        assert sorted(out) == sorted(expected_ids), 'chunk payloads missing after the exchange'
    return out


def merge_meshes(pieces):
    """``pieces``: list of (v [V,3] f32, f [T,3] i64, vkey [V] i64, axis [V] i8).  Vertices with
    the same (vkey, axis) are one vertex (seams between chunks / ranks).  Deterministic: output
    vertices ordered by (axis, vkey), faces in piece order; the representative of a merged vertex is its first
    occurrence.  On the GPU the grouping is a stable device radix sort of the lattice keys per axis
    (nksr_sort_pairs_u64_u32) + a flag scan; CPU tensors (the gloo tests) take the torch.unique route."""
    if all(len(p) > 4 and p[4] is not None for p in pieces):
        return _merge_flagged(pieces)
    v = torch.cat([p[0] for p in pieces])
    key = torch.cat([p[2] for p in pieces])
    ax = torch.cat([p[3] for p in pieces]).to(torch.int64)
    offs, faces = 0, []
    for p in pieces:
        faces.append(p[1] + offs)
        offs += p[0].shape[0]
    f = torch.cat(faces)
    new_index = torch.empty(v.shape[0], dtype=torch.int64, device=v.device)
    out_v, base = [], 0
    for a in range(3):
        sel = torch.nonzero(ax == a).reshape(-1)
        if sel.numel() == 0:
            continue
        if v.is_cuda:
            from . import ops
            ks, order = ops.sort_pairs(key[sel].contiguous(), torch.arange(sel.numel(), dtype=torch.int32, device=v.device))
            head = torch.ones(ks.numel(), dtype=torch.bool, device=v.device)
            head[1:] = ks[1:] != ks[:-1]
            rank_sorted = torch.cumsum(head.to(torch.int64), 0) - 1           # group id of every sorted position
            orig = sel[order.long()]                                          # stable sort: first of a group = first occurrence
            new_index[orig] = rank_sorted + base
            out_v.append(v[orig[head]])
            base += int(head.sum())
        else:
            uk, inv = torch.unique(key[sel], sorted=True, return_inverse=True)
            first = torch.full((uk.numel(),), v.shape[0], dtype=torch.int64, device=v.device)
            first.scatter_reduce_(0, inv, sel, reduce='amin')          # representative = first occurrence
            new_index[sel] = inv + base
            out_v.append(v[first])
            base += uk.numel()
    vv = torch.cat(out_v) if out_v else v[:0]
    return vv, new_index[f]


def _merge_flagged(pieces):
    """merge_meshes for pieces that say which of their vertices another piece may hold too (a fifth entry: uint8 flags,
    chunking.MultiChunkField.seam_flags -- 0.3 % of the vertices of the bench scene): the unflagged vertices pass through in piece
    order, the flagged ones are grouped by (axis, key) as in merge_meshes -- first occurrence the representative -- and follow them."""
    v = torch.cat([p[0] for p in pieces])
    flag = torch.cat([p[4].reshape(-1) for p in pieces]).to(torch.bool)
    offs, faces = 0, []
    for p in pieces:
        faces.append(p[1] + offs)
        offs += p[0].shape[0]
    f = torch.cat(faces)
    cand = torch.nonzero(flag).reshape(-1)
    n_int = v.shape[0] - cand.numel()
    new_index = torch.cumsum((~flag).to(torch.int64), 0) - 1            # (the entries of flagged vertices are overwritten below)
    out = [v[~flag]]
    if cand.numel():
        key = torch.cat([p[2] for p in pieces])[cand]
        ax = torch.cat([p[3] for p in pieces])[cand].to(torch.int64)
        cv, ci = merge_meshes([(v[cand], torch.arange(cand.numel(), dtype=torch.int64, device=v.device).view(-1, 1).expand(-1, 3), key, ax)])
        new_index[cand] = ci[:, 0] + n_int
        out.append(cv)
    return torch.cat(out), new_index[f]


def gather_meshes(v, f, vkey, axis, dst=0, seam=None):
    """Gathers the per-rank mesh pieces on rank ``dst`` ONLY (point-to-point, after one size collective); ``dst`` merges
    the seams and returns the full mesh, the other ranks keep their own piece.  ``seam`` (uint8 per vertex, or None on every rank):
    the vertices another rank may hold too -- rank ``dst`` then groups only those."""
    rank, ws = world()
    if not active():
        return v, f
    parts = [v.reshape(-1).contiguous(), f.reshape(-1).contiguous(), vkey.contiguous(), axis.to(torch.int8).contiguous()]
    if seam is not None:
        parts.append(seam.to(torch.uint8).contiguous())
    got = gather_tensors(parts, dst)
    if rank != dst:
        return v, f
    pieces = [(g[0].view(-1, 3), g[1].view(-1, 3), g[2], g[3]) + ((g[4],) if len(g) > 4 else ()) for g in got]
    return merge_meshes(pieces)


def merge_named(pieces):
    """``pieces``: list of (v [V,3] f32, f [T,3] i64, names [V,5] i64) -- the adaptive dual graph's mesh pieces, a vertex named by
    the ordered pair of primal cells its cube edge joins: (size A, key A, axis, size B, key B), the same on every rank
    (nksr_amd/meshing.py).  Vertices with the same name are one vertex.  Deterministic: output vertices in lexicographic name
    order -- the order of the single-process mesh --, faces in piece order, the representative of a merged vertex its first
    occurrence.  On the GPU four stable radix sorts (key B; axis and size B; key A; size A) + a flag scan; CPU tensors (the gloo
    tests) take torch.unique over the rows."""
    v = torch.cat([p[0] for p in pieces])
    names = torch.cat([p[2].reshape(-1, 5) for p in pieces])
    offs, faces = 0, []
    for p in pieces:
        faces.append(p[1] + offs)
        offs += p[0].shape[0]
    f = torch.cat(faces)
    n = v.shape[0]
    if n == 0:
        return v, f, names
    if v.is_cuda:
        from . import ops
        order = torch.arange(n, dtype=torch.int32, device=v.device)
        for col in (names[:, 4], names[:, 2] * 256 + names[:, 3], names[:, 1], names[:, 0]):       # least significant first; stable
            _, order = ops.sort_pairs(col[order.long()].contiguous(), order)
        srt = names[order.long()]
        head = torch.ones(n, dtype=torch.bool, device=v.device)
        head[1:] = (srt[1:] != srt[:-1]).any(1)
        group = torch.cumsum(head.to(torch.int64), 0) - 1
        new_index = torch.empty(n, dtype=torch.int64, device=v.device)
        new_index[order.long()] = group
        first = order.long()[head]                                     # stable sorts: the first of a group is its first occurrence
        return v[first], new_index[f], srt[head]
    un, inv = torch.unique(names, dim=0, sorted=True, return_inverse=True)
    first = torch.full((un.shape[0],), n, dtype=torch.int64)
    first.scatter_reduce_(0, inv, torch.arange(n, dtype=torch.int64), reduce='amin')
    return v[first], inv[f], un


def gather_named(v, f, names, dst=0):
    """gather_meshes for pieces whose vertices carry five-word names (merge_named): the pieces go to rank ``dst`` only, which
    merges the seams; the other ranks keep their own piece."""
    rank, ws = world()
    if not active():
        return v, f, names
    got = gather_tensors([v.reshape(-1).contiguous(), f.reshape(-1).contiguous(), names.reshape(-1).contiguous()], dst)
    if rank != dst:
        return v, f, names
    return merge_named([(a.view(-1, 3), b.view(-1, 3), c.view(-1, 5)) for a, b, c in got])
