"""csrc/chunks.hip against the oracle's statements (oracle/chunking.py): which points a chunk solves (core +- band), which
chunks weigh at a query and with what partition-of-unity weight (weight(): ((w up_x) dn_x) up_y ...), the translated
positions fl32(x + T_c) and the ordered blend  sum w f / max(sum w, 1e-20).  Integer and fp32 results bit for bit."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _cloud(rs, lo, hi, n):
    xyz = (rs.uniform(0, 1, (n, 3)) * (np.asarray(hi) - np.asarray(lo)) + np.asarray(lo)).astype(np.float32)
    return xyz


@pytest.mark.parametrize('grid_hi,cs,ov,band', [((9.7, 5.2, 1.0), 3.3, 0.4, 0.9), ((7.9, 7.9, 7.9), 2.7, 0.6, 1.1), ((4.0, 0.9, 0.8), 1.1, 0.2, 0.35)])
def test_membership_pairs_and_blend_match_the_oracle(grid_hi, cs, ov, band):
    from nksr_amd import chunking as pc, ops
    from nksr_amd._lib import call, ptr, stream
    from oracle import chunking as oc
    rs = np.random.RandomState(11)
    lo = [0.13, -0.4, 0.05]
    hi = [lo[a] + grid_hi[a] for a in range(3)]
    xyz = _cloud(rs, lo, hi, 20000)
    xyz[0], xyz[1] = np.float32(lo), np.float32(hi)                 # bounding box corners
    grid = oc.chunk_grid(lo, hi, cs)
    nchunk = grid[0] * grid[1] * grid[2]
    # points exactly on chunk faces and on the ends of the ramps
    for a in range(3):
        for j in range(1, grid[a]):
            for off in (0.0, -ov, ov, -band, band):
                p = xyz[rs.randint(len(xyz))].copy()
                p[a] = np.float32(lo[a] + j * cs + off)
                xyz = np.concatenate([xyz, p[None]])
    assert pc.chunk_grid(lo, hi, cs) == tuple(grid) or list(pc.chunk_grid(lo, hi, cs)) == list(grid)
    cores = {}
    for c in range(nchunk):
        cz, cy, cx = c % grid[2], (c // grid[2]) % grid[1], c // (grid[1] * grid[2])
        clo = [lo[0] + cx * cs, lo[1] + cy * cs, lo[2] + cz * cs]
        cores[c] = (clo, [clo[a] + cs for a in range(3)])
    xt = torch.from_numpy(xyz).to(DEV)
    # ---- membership of the solve: every other chunk wanted
    wanted = [c % 3 != 1 for c in range(nchunk)]
    idx, cid, counts = pc.select_chunk_points(xt, lo, grid, cs, band, wanted)
    idx, cid = idx.cpu().numpy(), cid.cpu().numpy()
    o = 0
    for c in range(nchunk):
        m = np.ones(len(xyz), bool)
        for a in range(3):
            if grid[a] > 1:
                m &= (xyz[:, a] >= oc._f32(cores[c][0][a] - band)) & (xyz[:, a] < oc._f32(cores[c][1][a] + band))
        want = np.nonzero(m)[0] if wanted[c] else np.zeros(0, np.int64)
        assert counts[c] == len(want), (c, counts[c], len(want))
        assert np.array_equal(idx[o:o + len(want)], want) and (cid[o:o + len(want)] == c).all()
        o += len(want)
    assert o == len(idx)
    # ---- blend pairs: all chunks but two present, random translations
    present = np.ones(nchunk, bool)
    present[rs.randint(nchunk)] = False
    shift = rs.uniform(-50, 50, (nchunk, 3)).astype(np.float32)
    flag = torch.from_numpy(np.where(present, 0, -1).astype(np.int32)).to(DEV)
    st = torch.from_numpy(shift).to(DEV)
    G, keep = pc.chunk_grid_struct(lo, grid, cs, None, ov, DEV, shift=st)
    n = len(xyz)
    cnt = torch.zeros(n + 1, dtype=torch.int32, device=DEV)
    call('nksr_chunk_pair_counts', C.byref(G), 1, ptr(xt), n, ptr(flag), ptr(cnt), stream())
    offs = ops.exclusive_sum_i32(cnt)
    m = int(offs[n])
    q = torch.empty(m, dtype=torch.int64, device=DEV)
    pcid = torch.empty(m, dtype=torch.int32, device=DEV)
    w = torch.empty(m, dtype=torch.float32, device=DEV)
    xq = torch.empty((m, 3), dtype=torch.float32, device=DEV)
    call('nksr_chunk_pair_fill', C.byref(G), 1, ptr(xt), n, ptr(flag), ptr(offs), ptr(q), ptr(pcid), ptr(w), ptr(xq), stream())
    import types
    cf = types.SimpleNamespace(cores=cores, ov=float(ov), grid=grid)          # what ChunkedField.weight reads
    W = np.stack([oc.ChunkedField.weight(cf, c, xyz) if present[c] else np.zeros(n, np.float32) for c in range(nchunk)], 1)        # [n, nchunk]
    qi, ci = np.nonzero(W > 0)                                              # row-major: ascending chunk per query = the fill order
    assert m == len(qi)
    assert np.array_equal(q.cpu().numpy(), qi) and np.array_equal(pcid.cpu().numpy(), ci)
    assert np.array_equal(w.cpu().numpy(), W[qi, ci])
    assert np.array_equal(xq.cpu().numpy(), (xyz[qi] + shift[ci]).astype(np.float32))
    assert np.array_equal(offs.cpu().numpy()[:-1], np.concatenate([[0], np.cumsum((W > 0).sum(1))])[:-1])
    # ---- the ordered blend
    f = rs.randn(m).astype(np.float32)
    g = rs.randn(m, 3).astype(np.float32)
    fo = torch.empty(n, dtype=torch.float32, device=DEV)
    go = torch.empty((n, 3), dtype=torch.float32, device=DEV)
    ft, gt = torch.from_numpy(f).to(DEV), torch.from_numpy(g).to(DEV)       # (named: a temporary would be freed before the launch reads it)
    call('nksr_chunk_blend', n, ptr(offs), ptr(w), ptr(ft), ptr(gt), ptr(fo), ptr(go), stream())
    num, den, gn = np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros((n, 3), np.float32)
    wv = W[qi, ci]
    for c in range(nchunk):                                                 # the oracle's loop: chunk after chunk
        s = np.nonzero(ci == c)[0]
        num[qi[s]] = (num[qi[s]] + f[s] * wv[s]).astype(np.float32)
        den[qi[s]] = (den[qi[s]] + wv[s]).astype(np.float32)
        gn[qi[s]] = (gn[qi[s]] + g[s] * wv[s, None]).astype(np.float32)
    den = np.maximum(den, np.float32(1e-20))
    assert np.array_equal(fo.cpu().numpy(), (num / den).astype(np.float32))
    assert np.array_equal(go.cpu().numpy(), (gn / den[:, None]).astype(np.float32))
    assert (den[(W > 0).sum(1) == 0] == np.float32(1e-20)).all() and (fo.cpu().numpy()[(W > 0).sum(1) == 0] == 0).all()


def test_chunk_kernels_reject_bad_arguments():
    from nksr_amd import _lib
    from nksr_amd._lib import ChunkGridT
    G = ChunkGridT()
    G.grid[0], G.grid[1], G.grid[2], G.reach = 2, 1, 1, 1
    x = torch.zeros((4, 3), device=DEV)
    cnt = torch.zeros(5, dtype=torch.int32, device=DEV)
    fl = torch.zeros(2, dtype=torch.int32, device=DEV)
    rc = _lib.lib.nksr_chunk_pair_counts(C.byref(G), 1, C.c_void_p(x.data_ptr()), C.c_int64(4), C.c_void_p(fl.data_ptr()), C.c_void_p(cnt.data_ptr()), None)
    assert rc != 0 and b'NULL' in _lib.lib.nksr_last_error()
