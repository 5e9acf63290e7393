"""ONE field, no chunk_size, through the matrix-free solve (the assembled system of such a cloud has nnz >= 2^31 and is refused:
fields/kernel_field.py assemble).  Reference call: examples/recons_waymo.py:30-37 (whole sequence in one reconstruct()).
python -m nksr_amd.tools.stress_single_field [points=8000000] [reps=2]"""
import sys
import time

import torch

import nksr_amd
from nksr_amd import utils


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    dev = torch.device('cuda:0')
    side = 80.0 * (n / 4e6) ** 0.5          # same point density as tools/stress_4m
    xyz, nrm = utils.synth_scene(n, seed=0, extent=(side, side, 10.0), n_objects=max(8, n // 125_000))
    xyz, nrm = torch.from_numpy(xyz).to(dev), torch.from_numpy(nrm).to(dev)
    rec = nksr_amd.Reconstructor(dev)
    for rep in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fld = rec.reconstruct(xyz, nrm, detail_level=1.0, fused_mode=True)
        mesh = fld.extract_dual_mesh(mise_iter=1)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print('rep %d: %.1f ms  %s' % (rep, dt * 1e3, {k: round(v * 1e3, 1) for k, v in rec.timing.items()}), flush=True)
    info = fld.solve_info
    f = mesh.f.long()
    V = mesh.v.shape[0]
    e = torch.cat([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
    key = torch.minimum(e[:, 0], e[:, 1]) * V + torch.maximum(e[:, 0], e[:, 1])
    _, cnt = torch.unique(key, return_counts=True)
    on = fld.evaluate_f(xyz[:: max(1, n // 200_000)].contiguous()).value.abs()
    print('N=%d M=%d stored entries of G,Q=%d iters=%d rel=%.2e alpha finite=%s  %.1f ms (%.1f M points/s)  V=%d F=%d open edges=%d non-manifold=%d  '
          'mean |f| at the input points=%.3e  peak mem %.1f GB' % (
              n, info['M'], fld.stored_entries() or 0, info['iters'], info['rel_residual'], bool(torch.isfinite(fld.alpha).all()), dt * 1e3, n / dt / 1e6,
              V, f.shape[0], int((cnt == 1).sum()), int((cnt > 2).sum()), float(on.mean()), torch.cuda.max_memory_allocated() / 1e9))


if __name__ == '__main__':
    main()
