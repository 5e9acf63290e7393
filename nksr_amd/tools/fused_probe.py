"""Matrix-free operator micro-benchmark on the bench workload: times nksr_fused_apply (sweep + gather)
with HIP events on the launch stream and compares it with the assembled CSR SpMV on the same system.
python -m nksr_amd.tools.fused_probe [points] [reps]"""
import sys

import torch

import nksr_amd
from nksr_amd import solver, utils


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    dev = torch.device('cuda:0')
    xyz, nrm = utils.synth_scene(n, seed=0)
    rec = nksr_amd.Reconstructor(dev)
    f = rec.reconstruct(torch.from_numpy(xyz).to(dev), torch.from_numpy(nrm).to(dev), detail_level=1.0, fused_mode=False)
    rowptr, cols, vals, diag = f.matrix
    M = rowptr.numel() - 1
    # the same system, matrix-free: rebuild the site sets exactly as Reconstructor._reconstruct_single does
    hp = rec.hparams
    from nksr_amd.nn.network import sort_cloud
    from nksr_amd.svh import inv_w0_f32
    xs = (torch.from_numpy(xyz).to(dev) * f.scale).contiguous()
    ks, xs, ns = sort_cloud(xs, torch.from_numpy(nrm).to(dev), inv_w0_f32(hp.voxel_size))
    nxyz = f.svh.get_voxel_centers(0)
    nval = torch.zeros_like(nxyz)
    op = f.fused_operator(xs, nxyz, nval, hp.solver.pos_weight / xs.shape[0], hp.solver.normal_weight / nxyz.shape[0] * hp.voxel_size ** 2,
                          pos_sorted_keys=ks, normal_sorted_keys=f.svh.level(0).keys)
    x = torch.randn(M, device=dev)
    y_csr = solver.spmv(rowptr, cols, vals, x)
    y_f = f.fused_apply(op, x)
    print('compact=%s rows_words=%d (dense slots %d)' % (op.get('compact'), op.get('rows_words', 0), 27 * f.svh.depth * op['rows_total']))
    print('M=%d rows=%d partial blocks=%d  max|y_fused - y_csr| / max|y| = %.3e' % (M, op['rows_total'], op['nblocks'],
                                                                                 float((y_f - y_csr).abs().max() / y_csr.abs().max())))
    for _ in range(3):
        f.fused_apply(op, x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f.fused_apply(op, x)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    slots = 27 * f.svh.depth * op['rows_total']
    phys = 4 * (op['rows_words'] if op.get('compact') else slots) + 4 * f.svh.depth * op['rows_total'] + 2 * 128 * op['nblocks'] + (128 + 3 * 108 + 12) * M      # csrc/fused.hip FusedOperator::bytes
    nnz = int(torch.count_nonzero(op['rows_all']).item())
    alg = 16.0 * nnz + 12 * M + 4
    print('fused apply: %.1f us  physical %.3f GB -> %.2f TB/s (%.1f%% of 8 TB/s);  SURVEY 8d figure (16 B x %d non-zero slots of %d) %.3f GB -> %.2f TB/s'
          % (ms * 1e3, phys / 1e9, phys / ms / 1e9, phys / ms / 1e9 / 8 * 100, nnz, slots, alg / 1e9, alg / ms / 1e9))


if __name__ == '__main__':
    main()
