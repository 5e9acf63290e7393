"""How compressible are the column indices of the bench matrix?  Span (max - min) of the columns inside
every 64-entry gather group of the logical CSR stream.  python -m nksr_amd.tools.col16_probe [points]"""
import sys

import torch

import nksr_amd
from nksr_amd import solver, utils


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    dev = torch.device('cuda:0')
    xyz, nrm = utils.synth_scene(n, seed=0)
    rec = nksr_amd.Reconstructor(dev)
    f = rec.reconstruct(torch.from_numpy(xyz).to(dev), torch.from_numpy(nrm).to(dev), detail_level=1.0)
    rowptr, cols, vals, diag = f.matrix
    nnz = f.nnz
    lc, _ = solver.csr_logical(rowptr, cols, vals)
    g = lc[: nnz // 64 * 64].view(-1, 64).long()
    span = g.max(1).values - g.min(1).values
    for bits in (8, 12, 16, 18, 20):
        print('groups with span < 2^%d: %.2f%%' % (bits, 100.0 * float((span < (1 << bits)).float().mean())))
    g16 = lc[: nnz // 16 * 16].view(-1, 16).long()
    s16 = g16.max(1).values - g16.min(1).values
    print('16-entry groups with span < 2^16: %.2f%%' % (100.0 * float((s16 < 65536).float().mean())))
    # delta to the previous entry (signed) fits in 16 bits?
    d = (lc[1:].long() - lc[:-1].long()).abs()
    print('|col[k] - col[k-1]| < 2^15: %.2f%%' % (100.0 * float((d < 32768).float().mean())))


if __name__ == '__main__':
    main()
