"""Largest single-field case that fits int32 nnz: 4 M points in ONE chunk (M ~ 4.5 M > 2^21, so the int32-column
CSR layout, col_format 0, is the one exercised at scale).  python -m nksr_amd.tools.stress_4m [points]"""
import sys
import time

import torch

import nksr_amd
from nksr_amd import solver, utils


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
    dev = torch.device('cuda:0')
    xyz, nrm = utils.synth_scene(n, seed=0, extent=(80.0, 80.0, 10.0), n_objects=32)
    xyz, nrm = torch.from_numpy(xyz).to(dev), torch.from_numpy(nrm).to(dev)
    rec = nksr_amd.Reconstructor(dev)
    rec.sync_timing = len(sys.argv) > 2
    for rep in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fld = rec.reconstruct(xyz, nrm, detail_level=1.0)
        mesh = fld.extract_dual_mesh(mise_iter=1)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print('rep %d: %.1f ms  %s' % (rep, dt * 1e3, {k: round(v * 1e3, 1) for k, v in rec.timing.items()}))
    info = fld.solve_info
    print({k: round(v * 1e3, 1) for k, v in rec.timing.items()})
    rowptr, cols, vals, diag = fld.matrix
    res = fld.rhs.double() - solver.spmv(rowptr, cols, vals, fld.alpha).double()
    rel = float(res.norm() / fld.rhs.double().norm())
    f = mesh.f.long()
    V = mesh.v.shape[0]
    e = torch.cat([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
    key = torch.minimum(e[:, 0], e[:, 1]) * V + torch.maximum(e[:, 0], e[:, 1])
    _, cnt = torch.unique(key, return_counts=True)
    print('N=%d M=%d nnz=%d col_format=%d iters=%d rel=%.2e independent residual=%.2e  %.1f ms (%.1f M points/s)  V=%d F=%d open edges=%d non-manifold=%d  peak mem %.1f GB' % (
        n, info['M'], info['nnz'], solver.col_format(cols), info['iters'], info['rel_residual'], rel, dt * 1e3, n / dt / 1e6, V, f.shape[0],
        int((cnt == 1).sum()), int((cnt > 2).sum()), torch.cuda.max_memory_allocated() / 1e9))


if __name__ == '__main__':
    main()
