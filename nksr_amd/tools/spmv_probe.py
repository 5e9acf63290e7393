"""SpMV micro-benchmark on the matrix of the bench workload: times kernel variants with HIP
events (torch events on the launch stream).  python -m nksr_amd.tools.spmv_probe [points]"""
import sys

import torch

import nksr_amd
from nksr_amd import solver, utils
from nksr_amd._lib import call, lib, ptr, stream


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    variants = [int(v) for v in sys.argv[2].split(',')] if len(sys.argv) > 2 else [0, 1]
    dev = torch.device('cuda:0')
    xyz, nrm = utils.synth_scene(n, seed=0)
    rec = nksr_amd.Reconstructor(dev)
    f = rec.reconstruct(torch.from_numpy(xyz).to(dev), torch.from_numpy(nrm).to(dev), detail_level=1.0, fused_mode=False)
    rowptr, cols, vals, diag = f.matrix
    M, nnz = rowptr.numel() - 1, f.nnz
    B = 8.0 * nnz + 12.0 * M + 4
    x = torch.randn(M, device=dev)
    y = torch.empty(M, device=dev)
    ws = torch.empty(int(lib.nksr_spmv_workspace_bytes(nnz)), dtype=torch.uint8, device=dev)
    fmt = solver.col_format(cols)
    call('nksr_spmv_plan', ptr(rowptr), M, nnz, fmt, ptr(ws), stream())
    print('M=%d nnz=%d nnz/row=%.1f bytes=%.3f GB col_format=%d' % (M, nnz, nnz / M, B / 1e9, fmt))
    y_ref = None
    for v in variants:
        call('nksr_spmv_set_variant', v)
        for _ in range(3):
            call('nksr_spmv_csr', ptr(rowptr), ptr(cols), ptr(vals), M, nnz, fmt, ptr(x), ptr(y), ptr(ws), stream())
        torch.cuda.synchronize()
        if v in (0, 2):      # real products: every variant must give the bits of variant 0 (same order of additions)
            if y_ref is None:
                y_ref = y.clone()
            else:
                print('variant %d vs variant %d: max |dy| = %.3e (max |y| = %.3e)' % (v, variants[0], float((y - y_ref).abs().max()), float(y_ref.abs().max())))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            call('nksr_spmv_csr', ptr(rowptr), ptr(cols), ptr(vals), M, nnz, fmt, ptr(x), ptr(y), ptr(ws), stream())
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print('variant %d: %.1f us  %.2f TB/s  (%.1f%% of 8 TB/s)' % (v, ms * 1e3, B / ms / 1e9, B / ms / 1e9 / 8 * 100))
    call('nksr_spmv_set_variant', 0)


if __name__ == '__main__':
    main()
