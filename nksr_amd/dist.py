"""Multi-GPU glue for the chunked path: one process per GPU,
``torch.distributed`` (backend "nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The solve of a chunk is independent of every other chunk (the reference runs chunks sequentially
on one device, examples/recons_by_chunk.py:26-29), so chunks are sharded over ranks with NO
collective on the solve path.  Exactly one exchange step precedes meshing -- every rank needs the
solved fields that overlap the cells it meshes -- and one gather step follows it:
  * exchange_payloads: all_gather of sizes, then all_gather of padded int64 / float32 buffers
    (payloads are tens of MB: latency-, not bandwidth-bound; the fully connected xGMI mesh
    serves an all_gather as direct peer copies, no ring bottleneck)
  * gather_meshes: same pattern to rank 0, then seam vertices are merged by their canonical
    (lattice key, axis) identity.
Everything here works on CPU tensors too, so the protocol is covered by world_size-2 gloo tests.
"""
import torch


def is_dist():
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


def world():
    import torch.distributed as dist
    return (dist.get_rank(), dist.get_world_size()) if is_dist() else (0, 1)


def partition_chunks(n_chunks, world_size, weights=None):
    """owner[c] for every chunk: greedy longest-processing-time balance on ``weights`` (point
    counts), ties broken by chunk index => identical on every rank."""
    if weights is None:
        return [c % world_size for c in range(n_chunks)]
    order = sorted(range(n_chunks), key=lambda c: (-int(weights[c]), c))
    load = [0] * world_size
    owner = [0] * n_chunks
    for c in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        owner[c] = r
        load[r] += int(weights[c])
    return owner


def _comm_device(t):
    import torch.distributed as dist
    return t.device if dist.get_backend() != 'gloo' else torch.device('cpu')


def all_gather_variable(t):
    """all_gather of 1-D tensors whose lengths differ per rank.  Returns the list of per-rank
    tensors (on t's device)."""
    import torch.distributed as dist
    rank, ws = world()
    if ws == 1:
        return [t]
    dev = _comm_device(t)
    n = torch.tensor([t.numel()], dtype=torch.int64, device=dev)
    sizes = [torch.zeros_like(n) for _ in range(ws)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    mx = max(max(sizes), 1)
    buf = torch.zeros(mx, dtype=t.dtype, device=dev)
    buf[:t.numel()] = t.to(dev)
    out = [torch.empty_like(buf) for _ in range(ws)]
    dist.all_gather(out, buf)
    return [o[:s].to(t.device) for o, s in zip(out, sizes)]


def exchange_payloads(local, expected_ids):
    """``local``: {chunk_id: (int64 tensor, float32 tensor)} for the chunks this rank owns.
    Returns the same dict for ALL chunks (``expected_ids``) on every rank."""
    rank, ws = world()
    if ws == 1:
        return dict(local)
    ids = sorted(local)
    if ids:
        dev = local[ids[0]][0].device
    else:                                  # idle rank (more ranks than chunks): still takes part in the collectives
        import torch.distributed as dist
        dev = torch.device('cpu') if dist.get_backend() == 'gloo' else torch.device('cuda', torch.cuda.current_device())
    head = torch.tensor([v for c in ids for v in (c, local[c][0].numel(), local[c][1].numel())], dtype=torch.int64, device=dev)
    ibuf = torch.cat([local[c][0].reshape(-1) for c in ids]) if ids else torch.zeros(0, dtype=torch.int64, device=dev)
    fbuf = torch.cat([local[c][1].reshape(-1) for c in ids]) if ids else torch.zeros(0, dtype=torch.float32, device=dev)
    heads, ibufs, fbufs = all_gather_variable(head), all_gather_variable(ibuf), all_gather_variable(fbuf)
    out = {}
    for h, ib, fb in zip(heads, ibufs, fbufs):
        io = fo = 0
        for k in range(0, h.numel(), 3):
            c, ni, nf = int(h[k]), int(h[k + 1]), int(h[k + 2])
            out[c] = (ib[io:io + ni], fb[fo:fo + nf])
            io += ni
            fo += nf
    assert sorted(out) == sorted(expected_ids), 'chunk payloads missing after the exchange'
    return out


def merge_meshes(pieces):
    """``pieces``: list of (v [V,3] f32, f [T,3] i64, vkey [V] i64, axis [V] i8).  Vertices with
    the same (vkey, axis) are one vertex (seams between chunks / ranks).  Deterministic: output
    vertices ordered by (axis, vkey), faces in piece order."""
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
        uk, inv = torch.unique(key[sel], sorted=True, return_inverse=True)
        first = torch.full((uk.numel(),), v.shape[0], dtype=torch.int64, device=v.device)
        first.scatter_reduce_(0, inv, sel, reduce='amin')          # representative = first occurrence
        new_index[sel] = inv + base
        out_v.append(v[first])
        base += uk.numel()
    vv = torch.cat(out_v) if out_v else v[:0]
    return vv, new_index[f]


def gather_meshes(v, f, vkey, axis, dst=0):
    """Gathers the per-rank mesh pieces; rank ``dst`` merges the seams and returns the full mesh,
    the other ranks keep their own piece (the merge is O(total mesh): doing it N times would cost
    weak-scaling efficiency for nothing)."""
    rank, ws = world()
    if ws == 1:
        return v, f
    vs = all_gather_variable(v.reshape(-1).contiguous())
    fs = all_gather_variable(f.reshape(-1).contiguous())
    ks = all_gather_variable(vkey.contiguous())
    as_ = all_gather_variable(axis.to(torch.int64).contiguous())
    if rank != dst:
        return v, f
    pieces = [(a.view(-1, 3), b.view(-1, 3), c, d.to(torch.int8)) for a, b, c, d in zip(vs, fs, ks, as_)]
    return merge_meshes(pieces)
