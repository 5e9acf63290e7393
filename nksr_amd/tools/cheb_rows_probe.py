"""Row lengths of the packed coarse-level block (what k_cheb16_step streams) on the bench scene.
python -m nksr_amd.tools.cheb_rows_probe [scene points]"""
import sys

import torch

import bench
import nksr_amd
from nksr_amd import configs
from nksr_amd.fields.kernel_field import KernelField


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    dev = torch.device('cuda:0')
    rec = nksr_amd.Reconstructor(dev, hparams=configs.get_hparams('ks', tree_depth=5))
    xyz, nrm, scale, owner, bounds, n_scene, _ = bench.terrain_setup(rec, dev, n, 0, 1)
    got = []
    orig = KernelField._coarse_precond

    def cap(self, *a, **k):
        r = orig(self, *a, **k)
        got.append((self, r))
        return r
    KernelField._coarse_precond = cap
    rec.reconstruct(xyz, nrm, detail_level=None, chunk_size=bench.TILE * scale, sharded_input=True, chunk_owner=owner, chunk_bounds=bounds)
    KernelField._coarse_precond = orig
    fld, pc = got[0]
    packed, prow = pc['keep'][0], pc['keep'][1]
    lens = (prow[1:] - prow[:-1]).long()
    nrow = lens.numel()
    print('coarse rows %d, kept entries %d (%.1f per row), assembled nnz %d, steps %d' % (nrow, int(lens.sum()), float(lens.float().mean()), pc['nnz'], pc['steps']))
    off = fld.svh.offsets
    c0 = pc['first_level']
    q = torch.tensor([0.1, 0.5, 0.9, 0.99, 1.0], device=dev)
    print('all rows: quantiles 10/50/90/99/100 %%: %s' % [int(v) for v in torch.quantile(lens.float(), q).tolist()])
    hist = torch.bincount((lens // 16).clamp(max=20))
    print('rows by ceil(len / 16) trips:', hist.tolist())


if __name__ == '__main__':
    main()
